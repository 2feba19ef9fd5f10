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
#define DVR_MAXK 16      /* largest num_dvr_units the restatement takes (dvrjanet.py:7; the CLI default is 3) */
#define GMP_DEGREE 5     /* models.py:26-28 builds GMP() with the defaults memory_length 11, degree 5 (gmp.py:6) */

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

static int64_t qat_param_count(const odpd_model_t* m);
static int64_t lstm_param_count(const odpd_model_t* m);
int64_t oracle_param_count(const odpd_model_t* m) {
    int64_t H = m->hidden, F = feat_dim(m->backbone);
    if (m->bits_w > 0 && (m->backbone == ODPD_LSTM || m->backbone == ODPD_VDLSTM)) return lstm_param_count(m);   /* quantised head(s) */
    if (m->bits_w > 0 && m->backbone != ODPD_DVRJANET && m->backbone != ODPD_DELTAJANET && m->backbone != ODPD_NEURALTX && m->backbone != ODPD_RVTDCNN && m->backbone != ODPD_PGJANET) return qat_param_count(m);   /* quantised models: + the quantiser scales */
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
        return 3 * (H * (H + 1) + H) + 2 * (H * 2 * H + H) + 2 * H + 2 + (m->bits_w > 0 ? 18 : 0);      /* + three scales per INT_Linear */
    case ODPD_GMP:      /* gmp.py:10-11: memory_length * (1 + (degree - 1) * memory_length); hidden = memory_length, degree 5 */
        return H * (1 + (GMP_DEGREE - 1) * H);
    case ODPD_DVRJANET: /* dvrjanet.py:13-30: cs (K = bits_w), seven HxH blocks, two H input columns, three H biases, two heads */
        return (m->bits_w > 0 && m->bits_w <= DVR_MAXK) ? m->bits_w + 7 * H * H + 7 * H + 2 : -1;
    case ODPD_BOJANET:  /* bojanet.py:15-26: two (6,16) FIR banks, two gates on the 12 envelopes (+bias) and the state, two heads; the phase
                           re-rotation (:41-53) cannot be built beyond hidden 18 */
        return H <= 18 ? 2 * H * H + 28 * H + 194 : -1;
    case ODPD_APNRRU:   /* apnrru.py:13-19, 45-53: two (3,16) FIR banks, C, Z (n), W_u (16, 8+n)+b, W_h (n,16)+b, two bias-free heads; n = 2H+3 */
        return 343 + 70 * H;
    case ODPD_MCLDNN:   /* mcldnn.py:21-27: conv2d_1 10C, conv1d 20C, conv2d_2 91, LSTM(5C -> 8) 160C + 320, fc 144 + 34 */
        return 190 * H + 589;
    case ODPD_DELTAJANET: /* deltajanet.py:96-111: two gates */
        return 2 * H * 6 + 2 * H * H + 4 * H + 2 * H + 2 + (m->bits_w > 0 ? 3 : 0);      /* + the INT_Linear head's scales */
    case ODPD_NEURALTX: /* neuraltx.py:18-38: two 5-tap FIRs, 4 -> C (bias), 4 depthwise k5, C -> 2, IQ_match (2,2); hidden = channels */
        return 10 + 4 * H + H + 4 * 5 * H + 2 * H + 4 + (m->bits_w > 0 ? 3 : 0);     /* + the INT_Linear IQ_match's scales */
    case ODPD_RVTDCNN:  /* rvtdcnn.py:19-33: Conv2d(1->3,k3) 27+3, fc_hid (H,36)+H, fc_out (2,H)+2; hidden = fc_hid_size (models.py:80-81) */
        return 30 + 36 * H + H + 2 * H + 2 + (m->bits_w > 0 ? 8 : 0);      /* + INT_Conv2D's two and the INT_Linears' three scales each */
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
/* LSTM family: lstm.py:45-48 (h0 = c0 = 0), vdlstm.py:56-81.  nn.LSTM gate order i,f,g,o.     */
/* ------------------------------------------------------------------------------------------ */
static inline real q_pow2(real scale) {   /* quantizers.py:56-65 */
    float l = rintf(log2f(fabsf((float)scale)));
    return (real)ldexp(1.0, (int)l);
}
/* returns q(x); *pass = 1 if x/s lies inside [Qn,Qp] (gradient passes), else 0 */
static inline real q_apply(real x, real s, int bits, real* pass) {
    real qn = -(real)(1 << (bits - 1)), qp = (real)((1 << (bits - 1)) - 1);
    real v = x / s;
    if (pass) *pass = (v >= qn && v <= qp) ? (real)1 : (real)0;
    v = v < qn ? qn : (v > qp ? qp : v);
    return (real)rint((double)v) * s;
}
/* An nn.Linear of a backbone the reference's surgery leaves otherwise alone (lstm, vdlstm: quant_envs.py:40-60, 290-306 swap only their
 * nn.Linear layers) as INT_Linear (quant_layers.py:48-85): y = q_w(W) q_a(in) + b on exact grid sums; the three scale parameters
 * (weight_quantizer, act_quantizer, out_quantizer) sit behind the layer's weight and bias in named_parameters() order and get zero
 * gradients (round() inside round_scale2pow2, quantizers.py:56-65). */
static void qlin_fwd(const real* p, int64_t o_w, int64_t o_b, int64_t o_q, int n_out, int n_in, int bw, int ba, const real* in,
                     real* inq, real* pass, real* out) {
    real sw = q_pow2(p[o_q]), sa = q_pow2(p[o_q + 1]);
    for (int j = 0; j < n_in; ++j) inq[j] = q_apply(in[j], sa, ba, &pass[j]);
    for (int o = 0; o < n_out; ++o) {
        double a = 0;
        for (int j = 0; j < n_in; ++j) a += (double)inq[j] * (double)q_apply(p[o_w + o * n_in + j], sw, bw, NULL);
        out[o] = (real)a + p[o_b + o];
    }
}
/* dp[w] += dout x q_a(in) through the weight quantiser's pass mask, dp[b] += dout, din += pass x q_w(W)^T dout */
static void qlin_bwd(const real* p, int64_t o_w, int64_t o_b, int64_t o_q, int n_out, int n_in, int bw, const real* dout,
                     const real* inq, const real* pass, real* dp, real* din) {
    real sw = q_pow2(p[o_q]);
    for (int o = 0; o < n_out; ++o) {
        dp[o_b + o] += dout[o];
        for (int j = 0; j < n_in; ++j) {
            real mk, wq = q_apply(p[o_w + o * n_in + j], sw, bw, &mk);
            dp[o_w + o * n_in + j] += dout[o] * inq[j] * mk;
            din[j] += pass[j] * dout[o] * wq;
        }
    }
}
typedef struct {
    int H, F, vd, q, bits_w, bits_a, eval;      /* q: the nn.Linear layers are INT_Linear (bits_w > 0); eval: ODPD_FLAG_EVAL */
    int64_t o_w_ih, o_w_hh, o_b_ih, o_b_hh, o_w_out, o_b_out, o_w_l1, o_b_l1, o_w_l2, o_b_l2, o_q_out, o_q_l1, o_q_l2, P;
} lstm_layout_t;
static void lstm_layout(const odpd_model_t* m, lstm_layout_t* g) {
    int64_t H = m->hidden, o = 0;
    g->H = (int)H; g->vd = (m->backbone == ODPD_VDLSTM); g->F = g->vd ? 4 : 2;
    g->q = m->bits_w > 0; g->bits_w = m->bits_w; g->bits_a = m->bits_a; g->eval = (m->flags & 1);
    g->o_q_out = g->o_q_l1 = g->o_q_l2 = -1;
    g->o_w_ih = o; o += 4 * H * g->F;
    g->o_w_hh = o; o += 4 * H * H;
    g->o_b_ih = o; o += 4 * H;
    g->o_b_hh = o; o += 4 * H;
    if (g->vd) {   /* fc_lambda_1, fc_lambda_2, fc_out (vdlstm.py:35-43) */
        g->o_w_l1 = o; o += 4 * H; g->o_b_l1 = o; o += 4; if (g->q) { g->o_q_l1 = o; o += 3; }
        g->o_w_l2 = o; o += 4 * H; g->o_b_l2 = o; o += 4; if (g->q) { g->o_q_l2 = o; o += 3; }
        g->o_w_out = o; o += 2 * 8; g->o_b_out = o; o += 2; if (g->q) { g->o_q_out = o; o += 3; }
    } else {
        g->o_w_l1 = g->o_b_l1 = g->o_w_l2 = g->o_b_l2 = 0;
        g->o_w_out = o; o += 2 * H; g->o_b_out = o; o += 2; if (g->q) { g->o_q_out = o; o += 3; }
    }
    g->P = o;
}
static int64_t lstm_param_count(const odpd_model_t* m) { lstm_layout_t L; lstm_layout(m, &L); return L.P; }
typedef struct {
    real xin[4], cw[4], sw[4], hp[MAXH], cp[MAXH], i[MAXH], f[MAXH], g[MAXH], o[MAXH], tc[MAXH], h[MAXH], l1[4], l2[4];
    real hq[MAXH], ph[MAXH], hq2[MAXH], ph2[MAXH], zq[8], pz[8];      /* quantised heads: q_a(inputs) and pass masks of the INT_Linear layers */
} lstm_step_t;

/* vdlstm.py:60-76: windows over the frame with CIRCULAR left padding: element k of window t is
 * sample (t - 3 + k) mod T */
static inline int vd_idx(int t, int k, int T) { int j = t - 3 + k; return ((j % T) + T) % T; }

static void lstm_seq_fwd(const lstm_layout_t* L, const real* p, int T, const real* x, real* y, lstm_step_t* S) {
    int H = L->H, F = L->F;
    real h[MAXH] = {0}, c[MAXH] = {0};
    lstm_step_t tmp;
    for (int t = 0; t < T; ++t) {
        lstm_step_t* s = S ? &S[t] : &tmp;
        if (L->vd) {
            for (int k = 0; k < 4; ++k) {
                int j = vd_idx(t, k, T);
                real I = x[2 * j], Q = x[2 * j + 1], a = (real)sqrt((double)(I * I + Q * Q));
                s->xin[k] = a; s->cw[k] = I / a; s->sw[k] = Q / a;
            }
        } else { s->xin[0] = x[2 * t]; s->xin[1] = x[2 * t + 1]; }
        for (int j = 0; j < H; ++j) { s->hp[j] = h[j]; s->cp[j] = c[j]; }
        for (int j = 0; j < H; ++j) {
            real pre[4];
            for (int k = 0; k < 4; ++k) {
                real a = p[L->o_b_ih + k * H + j], b = p[L->o_b_hh + k * H + j];
                for (int i = 0; i < F; ++i) a += p[L->o_w_ih + (k * H + j) * F + i] * s->xin[i];
                for (int i = 0; i < H; ++i) b += p[L->o_w_hh + (k * H + j) * H + i] * s->hp[i];
                pre[k] = a + b;
            }
            s->i[j] = sigm(pre[0]); s->f[j] = sigm(pre[1]); s->g[j] = tanhr(pre[2]); s->o[j] = sigm(pre[3]);
            c[j] = s->f[j] * s->cp[j] + s->i[j] * s->g[j];
            s->tc[j] = tanhr(c[j]);
            h[j] = s->o[j] * s->tc[j];
            s->h[j] = h[j];
        }
        if (L->q) {      /* the heads as INT_Linear; fc_out's 16-bit output quantiser in eval mode only (quant_layers.py:77-80) */
            real z[8];
            if (L->vd) {
                qlin_fwd(p, L->o_w_l1, L->o_b_l1, L->o_q_l1, 4, H, L->bits_w, L->bits_a, h, s->hq, s->ph, s->l1);
                qlin_fwd(p, L->o_w_l2, L->o_b_l2, L->o_q_l2, 4, H, L->bits_w, L->bits_a, h, s->hq2, s->ph2, s->l2);
                for (int k = 0; k < 4; ++k) { z[k] = s->l1[k] * s->cw[k]; z[4 + k] = s->l2[k] * s->sw[k]; }
                qlin_fwd(p, L->o_w_out, L->o_b_out, L->o_q_out, 2, 8, L->bits_w, L->bits_a, z, s->zq, s->pz, &y[2 * t]);
            } else {
                qlin_fwd(p, L->o_w_out, L->o_b_out, L->o_q_out, 2, H, L->bits_w, L->bits_a, h, s->hq, s->ph, &y[2 * t]);
            }
            if (L->eval) for (int cc = 0; cc < 2; ++cc) y[2 * t + cc] = q_apply(y[2 * t + cc], q_pow2(p[L->o_q_out + 2]), 16, NULL);
        } else if (L->vd) {   /* vdlstm.py:78-80 */
            real z[8];
            for (int k = 0; k < 4; ++k) {
                real a = p[L->o_b_l1 + k], b = p[L->o_b_l2 + k];
                for (int j = 0; j < H; ++j) { a += p[L->o_w_l1 + k * H + j] * h[j]; b += p[L->o_w_l2 + k * H + j] * h[j]; }
                s->l1[k] = a; s->l2[k] = b;
                z[k] = a * s->cw[k]; z[4 + k] = b * s->sw[k];
            }
            for (int cc = 0; cc < 2; ++cc) {
                real a = p[L->o_b_out + cc];
                for (int k = 0; k < 8; ++k) a += p[L->o_w_out + cc * 8 + k] * z[k];
                y[2 * t + cc] = a;
            }
        } else {
            for (int cc = 0; cc < 2; ++cc) {
                real a = p[L->o_b_out + cc];
                for (int j = 0; j < H; ++j) a += p[L->o_w_out + cc * H + j] * h[j];
                y[2 * t + cc] = a;
            }
        }
    }
}
static void lstm_seq_bwd(const lstm_layout_t* L, const real* p, int T, const real* x, const real* dy, const lstm_step_t* S,
                         real* dp, real* dx) {
    int H = L->H, F = L->F;
    real dh[MAXH] = {0}, dc[MAXH] = {0};
    if (dx) memset(dx, 0, sizeof(real) * 2 * T);
    for (int t = T - 1; t >= 0; --t) {
        const lstm_step_t* s = &S[t];
        real dht[MAXH], dxin[4] = {0}, dcw[4] = {0}, dsw[4] = {0};
        for (int j = 0; j < H; ++j) dht[j] = dh[j];
        if (L->q) {
            if (L->vd) {
                real dz[8] = {0}, d1[4], d2[4];
                qlin_bwd(p, L->o_w_out, L->o_b_out, L->o_q_out, 2, 8, L->bits_w, &dy[2 * t], s->zq, s->pz, dp, dz);
                for (int k = 0; k < 4; ++k) {
                    d1[k] = dz[k] * s->cw[k]; d2[k] = dz[4 + k] * s->sw[k];
                    dcw[k] = dz[k] * s->l1[k]; dsw[k] = dz[4 + k] * s->l2[k];
                }
                qlin_bwd(p, L->o_w_l1, L->o_b_l1, L->o_q_l1, 4, H, L->bits_w, d1, s->hq, s->ph, dp, dht);
                qlin_bwd(p, L->o_w_l2, L->o_b_l2, L->o_q_l2, 4, H, L->bits_w, d2, s->hq2, s->ph2, dp, dht);
            } else {
                qlin_bwd(p, L->o_w_out, L->o_b_out, L->o_q_out, 2, H, L->bits_w, &dy[2 * t], s->hq, s->ph, dp, dht);
            }
        } else if (L->vd) {
            real dz[8] = {0};
            for (int cc = 0; cc < 2; ++cc) {
                real d = dy[2 * t + cc];
                dp[L->o_b_out + cc] += d;
                for (int k = 0; k < 8; ++k) {
                    real zk = k < 4 ? s->l1[k] * s->cw[k] : s->l2[k - 4] * s->sw[k - 4];
                    dp[L->o_w_out + cc * 8 + k] += d * zk;
                    dz[k] += d * p[L->o_w_out + cc * 8 + k];
                }
            }
            for (int k = 0; k < 4; ++k) {
                real d1 = dz[k] * s->cw[k], d2 = dz[4 + k] * s->sw[k];
                dcw[k] = dz[k] * s->l1[k]; dsw[k] = dz[4 + k] * s->l2[k];
                dp[L->o_b_l1 + k] += d1; dp[L->o_b_l2 + k] += d2;
                for (int j = 0; j < H; ++j) {
                    dp[L->o_w_l1 + k * H + j] += d1 * s->h[j]; dp[L->o_w_l2 + k * H + j] += d2 * s->h[j];
                    dht[j] += d1 * p[L->o_w_l1 + k * H + j] + d2 * p[L->o_w_l2 + k * H + j];
                }
            }
        } else {
            for (int cc = 0; cc < 2; ++cc) {
                real d = dy[2 * t + cc];
                dp[L->o_b_out + cc] += d;
                for (int j = 0; j < H; ++j) { dp[L->o_w_out + cc * H + j] += d * s->h[j]; dht[j] += d * p[L->o_w_out + cc * H + j]; }
            }
        }
        real dhp[MAXH] = {0}, dcp[MAXH];
        for (int j = 0; j < H; ++j) {
            real dO = dht[j] * s->tc[j];
            real dct = dc[j] + dht[j] * s->o[j] * ((real)1 - s->tc[j] * s->tc[j]);
            real dI = dct * s->g[j], dF = dct * s->cp[j], dG = dct * s->i[j];
            dcp[j] = dct * s->f[j];
            real dpre[4] = {dI * s->i[j] * ((real)1 - s->i[j]), dF * s->f[j] * ((real)1 - s->f[j]),
                            dG * ((real)1 - s->g[j] * s->g[j]), dO * s->o[j] * ((real)1 - s->o[j])};
            for (int k = 0; k < 4; ++k) {
                dp[L->o_b_ih + k * H + j] += dpre[k]; dp[L->o_b_hh + k * H + j] += dpre[k];
                for (int i = 0; i < F; ++i) { dp[L->o_w_ih + (k * H + j) * F + i] += dpre[k] * s->xin[i]; dxin[i] += dpre[k] * p[L->o_w_ih + (k * H + j) * F + i]; }
                for (int i = 0; i < H; ++i) { dp[L->o_w_hh + (k * H + j) * H + i] += dpre[k] * s->hp[i]; dhp[i] += dpre[k] * p[L->o_w_hh + (k * H + j) * H + i]; }
            }
        }
        for (int j = 0; j < H; ++j) { dh[j] = dhp[j]; dc[j] = dcp[j]; }
        if (dx) {
            if (L->vd) {
                for (int k = 0; k < 4; ++k) {
                    int j = vd_idx(t, k, T);
                    real I = x[2 * j], Q = x[2 * j + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
                    real da = dxin[k] - dcw[k] * I / a2 - dsw[k] * Q / a2;
                    dx[2 * j] += dcw[k] / a + da * I / a;
                    dx[2 * j + 1] += dsw[k] / a + da * Q / a;
                }
            } else { dx[2 * t] += dxin[0]; dx[2 * t + 1] += dxin[1]; }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Delta-GRU family: deltagru.py:59-77,211-264 and deltagru_tcnskip.py:87-103,248-293           */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int H, tres;
    int64_t o_w_ih, o_w_hh, o_b_ih, o_b_hh, o_w_out, o_b_out, o_tcn0, o_tcn2;
} delta_layout_t;
static void delta_layout(const odpd_model_t* m, delta_layout_t* g) {
    int64_t H = m->hidden, o = 0;
    g->H = (int)H; g->tres = (m->backbone == ODPD_TRES_DELTAGRU);
    g->o_w_ih = o; o += 3 * H * 6;
    g->o_w_hh = o; o += 3 * H * H;
    g->o_b_ih = g->o_b_hh = g->o_b_out = g->o_tcn0 = g->o_tcn2 = -1;
    if (!g->tres) { g->o_b_ih = o; o += 3 * H; g->o_b_hh = o; o += 3 * H; }
    g->o_w_out = o; o += 2 * H;
    if (!g->tres) { g->o_b_out = o; o += 2; }
    else { g->o_tcn0 = o; o += 18; g->o_tcn2 = o; o += 6; }
}
static inline real hswish(real v) { real r = v + (real)3; r = r < 0 ? 0 : (r > (real)6 ? (real)6 : r); return v * r / (real)6; }
static inline real hswish_grad(real v) { return v < (real)-3 ? (real)0 : (v <= (real)3 ? v / (real)3 + (real)0.5 : (real)1); }

typedef struct {
    real f[6], dxm[6], mx[6], hprev[MAXH], dhm[MAXH], mh[MAXH], r[MAXH], z[MAXH], n[MAXH], dmnh[MAXH], h[MAXH];
    real s1[3], s2[2];  /* TCN pre-activations (tres) */
} delta_step_t;

static void delta_feat(const delta_layout_t* L, const real* x, int t, int T, real* f) {
    real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
    f[0] = I; f[1] = Q; f[2] = a; f[3] = a * a * a;
    if (L->tres) { int tn = (t + 1) % T; f[4] = x[2 * tn]; f[5] = x[2 * tn + 1]; }   /* torch.roll(x, -1) */
    else { f[4] = Q / a; f[5] = I / a; }
}

static void delta_seq_fwd(const odpd_model_t* m, const delta_layout_t* L, const real* p, int T, const real* x, real* y,
                          delta_step_t* S, double* stats) {
    int H = L->H;
    const real thx = (real)(float)m->thx, thh = (real)(float)m->thh;
    real xp[6] = {0}, h[MAXH] = {0}, hp[MAXH] = {0}, dm[3 * MAXH], dmnh[MAXH];
    for (int j = 0; j < H; ++j) {
        if (L->tres) { dm[j] = dm[H + j] = dm[2 * H + j] = 0; dmnh[j] = 0; }
        else {   /* deltagru.py:165-170 */
            dm[j] = p[L->o_b_ih + j] + p[L->o_b_hh + j];
            dm[H + j] = p[L->o_b_ih + H + j] + p[L->o_b_hh + H + j];
            dm[2 * H + j] = p[L->o_b_ih + 2 * H + j];
            dmnh[j] = p[L->o_b_hh + 2 * H + j];
        }
    }
    delta_step_t tmp;
    double zx = 0, zh = 0;
    for (int t = 0; t < T; ++t) {
        delta_step_t* s = S ? &S[t] : &tmp;
        delta_feat(L, x, t, T, s->f);
        for (int i = 0; i < 6; ++i) {
            real d = s->f[i] - xp[i], ad = (real)fabs((double)d);
            s->mx[i] = (ad < thx) ? (real)0 : (real)1;
            s->dxm[i] = s->mx[i] != 0 ? d : (real)0;
            if (s->dxm[i] == 0) zx += 1;
            if (ad >= thx) xp[i] = s->f[i];
        }
        for (int j = 0; j < H; ++j) {
            real d = h[j] - hp[j], ad = (real)fabs((double)d);
            s->hprev[j] = h[j];
            s->mh[j] = (ad < thh) ? (real)0 : (real)1;
            s->dhm[j] = s->mh[j] != 0 ? d : (real)0;
            if (s->dhm[j] == 0) zh += 1;
            if (ad >= thh) hp[j] = h[j];
        }
        for (int j = 0; j < H; ++j) {
            real mx[3], mh[3];
            for (int k = 0; k < 3; ++k) {
                real a = 0, b = 0;
                for (int i = 0; i < 6; ++i) a += p[L->o_w_ih + (k * H + j) * 6 + i] * s->dxm[i];
                for (int i = 0; i < H; ++i) b += p[L->o_w_hh + (k * H + j) * H + i] * s->dhm[i];
                mx[k] = a + dm[k * H + j]; mh[k] = b;
            }
            dm[j] = mx[0] + mh[0]; dm[H + j] = mx[1] + mh[1]; dm[2 * H + j] = mx[2];
            dmnh[j] = mh[2] + dmnh[j];
            s->r[j] = sigm(dm[j]); s->z[j] = sigm(dm[H + j]); s->dmnh[j] = dmnh[j];
            s->n[j] = tanhr(dm[2 * H + j] + s->r[j] * dmnh[j]);
        }
        for (int j = 0; j < H; ++j) { h[j] = ((real)1 - s->z[j]) * s->n[j] + s->z[j] * h[j]; s->h[j] = h[j]; }
        for (int c = 0; c < 2; ++c) {
            real a = L->tres ? (real)0 : p[L->o_b_out + c];
            for (int j = 0; j < H; ++j) a += p[L->o_w_out + c * H + j] * h[j];
            y[2 * t + c] = a;
        }
        if (L->tres) {   /* deltagru_tcnskip.py:32-49,88,102: conv(2->3,k3,dil16,pad16) HS conv(3->2,k1) HS */
            for (int c = 0; c < 3; ++c) {
                real a = 0;
                for (int i = 0; i < 2; ++i)
                    for (int k = 0; k < 3; ++k) {
                        int tt = t + 16 * (k - 1);
                        if (tt >= 0 && tt < T) a += p[L->o_tcn0 + (c * 2 + i) * 3 + k] * x[2 * tt + i];
                    }
                s->s1[c] = a;
            }
            for (int o = 0; o < 2; ++o) {
                real a = 0;
                for (int c = 0; c < 3; ++c) a += p[L->o_tcn2 + o * 3 + c] * hswish(s->s1[c]);
                s->s2[o] = a;
                y[2 * t + o] += hswish(a);
            }
        }
    }
    if (stats) { stats[0] += zx; stats[1] += 6.0 * T; stats[2] += zh; stats[3] += (double)H * T; }
}

static void delta_seq_bwd(const odpd_model_t* m, const delta_layout_t* L, const real* p, int T, const real* x, const real* dy,
                          const delta_step_t* S, real* dp, real* dx) {
    int H = L->H;
    real Gh[MAXH] = {0}, Ghp[MAXH] = {0}, Gxp[6] = {0}, Gdm[3 * MAXH] = {0}, Gnh[MAXH] = {0};
    real* dfeat = (real*)calloc((size_t)T * 6, sizeof(real));
    if (dx) memset(dx, 0, sizeof(real) * 2 * T);
    for (int t = T - 1; t >= 0; --t) {
        const delta_step_t* s = &S[t];
        for (int c = 0; c < 2; ++c) {
            real d = dy[2 * t + c];
            if (!L->tres) dp[L->o_b_out + c] += d;
            for (int j = 0; j < H; ++j) { dp[L->o_w_out + c * H + j] += d * s->h[j]; Gh[j] += d * p[L->o_w_out + c * H + j]; }
        }
        if (L->tres) {
            real dh1[3] = {0};
            for (int o = 0; o < 2; ++o) {
                real d2 = dy[2 * t + o] * hswish_grad(s->s2[o]);
                for (int c = 0; c < 3; ++c) { dp[L->o_tcn2 + o * 3 + c] += d2 * hswish(s->s1[c]); dh1[c] += d2 * p[L->o_tcn2 + o * 3 + c]; }
            }
            for (int c = 0; c < 3; ++c) {
                real d1 = dh1[c] * hswish_grad(s->s1[c]);
                for (int i = 0; i < 2; ++i)
                    for (int k = 0; k < 3; ++k) {
                        int tt = t + 16 * (k - 1);
                        if (tt >= 0 && tt < T) {
                            dp[L->o_tcn0 + (c * 2 + i) * 3 + k] += d1 * x[2 * tt + i];
                            if (dx) dx[2 * tt + i] += d1 * p[L->o_tcn0 + (c * 2 + i) * 3 + k];
                        }
                    }
            }
        }
        real Ghprev[MAXH];
        for (int j = 0; j < H; ++j) {
            real dn = Gh[j] * ((real)1 - s->z[j]), dz = Gh[j] * (s->hprev[j] - s->n[j]);
            Ghprev[j] = Gh[j] * s->z[j];
            real dpre = dn * ((real)1 - s->n[j] * s->n[j]);
            Gdm[2 * H + j] += dpre;
            real dr = dpre * s->dmnh[j];
            Gnh[j] += dpre * s->r[j];
            Gdm[j] += dr * s->r[j] * ((real)1 - s->r[j]);
            Gdm[H + j] += dz * s->z[j] * ((real)1 - s->z[j]);
        }
        real ddx[6] = {0}, ddh[MAXH] = {0};
        for (int j = 0; j < H; ++j)
            for (int k = 0; k < 3; ++k) {
                real gx = Gdm[k * H + j], gh = (k < 2) ? Gdm[k * H + j] : Gnh[j];
                for (int i = 0; i < 6; ++i) { dp[L->o_w_ih + (k * H + j) * 6 + i] += gx * s->dxm[i]; ddx[i] += gx * p[L->o_w_ih + (k * H + j) * 6 + i]; }
                for (int i = 0; i < H; ++i) { dp[L->o_w_hh + (k * H + j) * H + i] += gh * s->dhm[i]; ddh[i] += gh * p[L->o_w_hh + (k * H + j) * H + i]; }
            }
        for (int i = 0; i < 6; ++i) {
            real mk = s->mx[i];
            dfeat[t * 6 + i] += mk * ddx[i] + mk * Gxp[i];
            Gxp[i] = ((real)1 - mk) * Gxp[i] - mk * ddx[i];
        }
        for (int j = 0; j < H; ++j) {
            real mk = s->mh[j];
            Ghprev[j] += mk * ddh[j] + mk * Ghp[j];
            Ghp[j] = ((real)1 - mk) * Ghp[j] - mk * ddh[j];
            Gh[j] = Ghprev[j];
        }
    }
    if (!L->tres)
        for (int j = 0; j < H; ++j) {
            dp[L->o_b_ih + j] += Gdm[j]; dp[L->o_b_hh + j] += Gdm[j];
            dp[L->o_b_ih + H + j] += Gdm[H + j]; dp[L->o_b_hh + H + j] += Gdm[H + j];
            dp[L->o_b_ih + 2 * H + j] += Gdm[2 * H + j]; dp[L->o_b_hh + 2 * H + j] += Gnh[j];
        }
    if (dx)
        for (int t = 0; t < T; ++t) {
            const real* df = dfeat + t * 6;
            real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
            if (L->tres) {
                real da = df[2] + (real)3 * a * a * df[3];
                dx[2 * t] += df[0] + da * I / a; dx[2 * t + 1] += df[1] + da * Q / a;
                int tn = (t + 1) % T;
                dx[2 * tn] += df[4]; dx[2 * tn + 1] += df[5];
            } else {
                real da = df[2] + (real)3 * a * a * df[3] - (Q / a2) * df[4] - (I / a2) * df[5];
                dx[2 * t] += df[0] + df[5] / a + da * I / a; dx[2 * t + 1] += df[1] + df[4] / a + da * Q / a;
            }
        }
    free(dfeat);
}

/* ------------------------------------------------------------------------------------------ */
/* DeltaJANET: deltajanet.py:50-64 (features [I,Q,a,a^3,sin,cos], fc_out with bias) around DeltaJANETLayer (:67-274): two
 * gates [f; g] of H rows each (weight_ih_l0 (2H,6), weight_hh_l0 (2H,H), bias_ih_l0, bias_hh_l0); accumulator dm (2H) starts
 * at bias_ih + bias_hh (:162-166); per step (:229-251) dx = x - x_p, dh = h - h_p (masked below the thresholds, which the
 * wrapper fixes at 0: DeltaJANET builds its layer with thx = thh = 0, :23-27 — nothing is ever masked), x_p <- x, h_p <- h,
 * dm = (dx W_ih^T + dm) + dh W_hh^T (:198-206), f = sigmoid(dm_f), g = sigmoid(dm_g) (:246-247: the candidate is a sigmoid too),
 * h = (1 - f) g + f h (:250).  Counters num_d{x,h}_{zeros,numel} as deltagru.  Parameter order: rnn.weight_ih_l0,
 * rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, fc_out.weight, fc_out.bias. */
/* ------------------------------------------------------------------------------------------ */
/* `--quant` (bits_w > 0): the surgery finds one nn.Linear, fc_out (the cell's gates are nn.Parameter tensors, deltajanet.py:100-113),
 * and makes it an INT_Linear (qlin_fwd / qlin_bwd above; three scale parameters behind fc_out.bias; ODPD_FLAG_EVAL: 16-bit output grid). */
typedef struct { int H, q, bits_w, bits_a, eval; int64_t o_w_ih, o_w_hh, o_b_ih, o_b_hh, o_w_out, o_b_out, o_q_out, P; } dj_layout_t;
static void dj_layout(const odpd_model_t* m, dj_layout_t* g) {
    int64_t H = m->hidden, o = 0;
    g->H = (int)H;
    g->q = m->bits_w > 0; g->bits_w = m->bits_w; g->bits_a = m->bits_a; g->eval = (m->flags & 1);
    g->o_w_ih = o; o += 2 * H * 6; g->o_w_hh = o; o += 2 * H * H;
    g->o_b_ih = o; o += 2 * H; g->o_b_hh = o; o += 2 * H;
    g->o_w_out = o; o += 2 * H; g->o_b_out = o; o += 2;
    g->o_q_out = -1;
    if (g->q) { g->o_q_out = o; o += 3; }
    g->P = o;
}
typedef struct { real f[6], dxm[6], hprev[MAXH], dhm[MAXH], fg[MAXH], gg[MAXH], h[MAXH], hq[MAXH], ph[MAXH]; } dj_step_t;
static void dj_seq_fwd(const dj_layout_t* L, const real* p, int T, const real* x, real* y, dj_step_t* S, double* stats) {
    const int H = L->H;
    real xp[6] = {0}, h[MAXH] = {0}, hp[MAXH] = {0}, dm[2 * MAXH];
    for (int j = 0; j < 2 * H; ++j) dm[j] = p[L->o_b_ih + j] + p[L->o_b_hh + j];
    dj_step_t tmp;
    double zx = 0, zh = 0;
    for (int t = 0; t < T; ++t) {
        dj_step_t* s = S ? &S[t] : &tmp;
        const real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
        s->f[0] = I; s->f[1] = Q; s->f[2] = a; s->f[3] = a * a * a; s->f[4] = Q / a; s->f[5] = I / a;
        for (int i = 0; i < 6; ++i) { s->dxm[i] = s->f[i] - xp[i]; if (s->dxm[i] == 0) zx += 1; xp[i] = s->f[i]; }
        for (int j = 0; j < H; ++j) { s->hprev[j] = h[j]; s->dhm[j] = h[j] - hp[j]; if (s->dhm[j] == 0) zh += 1; hp[j] = h[j]; }
        for (int j = 0; j < 2 * H; ++j) {
            real ax = 0, ah = 0;
            for (int i = 0; i < 6; ++i) ax += p[L->o_w_ih + j * 6 + i] * s->dxm[i];
            for (int i = 0; i < H; ++i) ah += p[L->o_w_hh + j * H + i] * s->dhm[i];
            dm[j] = (ax + dm[j]) + ah;
        }
        for (int j = 0; j < H; ++j) {
            s->fg[j] = sigm(dm[j]); s->gg[j] = sigm(dm[H + j]);
            h[j] = ((real)1 - s->fg[j]) * s->gg[j] + s->fg[j] * h[j];
            s->h[j] = h[j];
        }
        if (L->q) {
            qlin_fwd(p, L->o_w_out, L->o_b_out, L->o_q_out, 2, H, L->bits_w, L->bits_a, h, s->hq, s->ph, &y[2 * t]);
            if (L->eval) for (int c = 0; c < 2; ++c) y[2 * t + c] = q_apply(y[2 * t + c], q_pow2(p[L->o_q_out + 2]), 16, NULL);
        } else for (int c = 0; c < 2; ++c) {
            real acc = p[L->o_b_out + c];
            for (int j = 0; j < H; ++j) acc += p[L->o_w_out + c * H + j] * h[j];
            y[2 * t + c] = acc;
        }
    }
    if (stats) { stats[0] += zx; stats[1] += 6.0 * T; stats[2] += zh; stats[3] += (double)H * T; }
}
/* back-propagation with carried accumulator gradients (d dm_t feeds every later step through the running sum) */
static void dj_seq_bwd(const dj_layout_t* L, const real* p, int T, const real* x, const real* dy, const dj_step_t* S, real* dp, real* dx) {
    const int H = L->H;
    real Gh[MAXH] = {0}, Ghp[MAXH] = {0}, Gxp[6] = {0}, Gdm[2 * MAXH] = {0};
    for (int t = T - 1; t >= 0; --t) {
        const dj_step_t* s = &S[t];
        if (L->q) qlin_bwd(p, L->o_w_out, L->o_b_out, L->o_q_out, 2, H, L->bits_w, &dy[2 * t], s->hq, s->ph, dp, Gh);
        else for (int c = 0; c < 2; ++c) {
            const real d = dy[2 * t + c];
            dp[L->o_b_out + c] += d;
            for (int j = 0; j < H; ++j) { dp[L->o_w_out + c * H + j] += d * s->h[j]; Gh[j] += d * p[L->o_w_out + c * H + j]; }
        }
        real Ghprev[MAXH];
        for (int j = 0; j < H; ++j) {
            const real dg = Gh[j] * ((real)1 - s->fg[j]), df = Gh[j] * (s->hprev[j] - s->gg[j]);
            Ghprev[j] = Gh[j] * s->fg[j];
            Gdm[j] += df * s->fg[j] * ((real)1 - s->fg[j]);
            Gdm[H + j] += dg * s->gg[j] * ((real)1 - s->gg[j]);
        }
        real ddx[6] = {0}, ddh[MAXH] = {0};
        for (int j = 0; j < 2 * H; ++j) {
            const real g = Gdm[j];
            for (int i = 0; i < 6; ++i) { dp[L->o_w_ih + j * 6 + i] += g * s->dxm[i]; ddx[i] += g * p[L->o_w_ih + j * 6 + i]; }
            for (int i = 0; i < H; ++i) { dp[L->o_w_hh + j * H + i] += g * s->dhm[i]; ddh[i] += g * p[L->o_w_hh + j * H + i]; }
        }
        if (dx) {   /* dx_t = f_t - f_{t-1}: dL/df_t = ddx_t - ddx_{t+1} (Gxp carries -ddx_{t+1}) */
            real df[6];
            for (int i = 0; i < 6; ++i) { df[i] = ddx[i] + Gxp[i]; Gxp[i] = -ddx[i]; }
            const real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
            const real da = df[2] + (real)3 * a * a * df[3] - (Q / a2) * df[4] - (I / a2) * df[5];
            dx[2 * t] = df[0] + df[5] / a + da * I / a; dx[2 * t + 1] = df[1] + df[4] / a + da * Q / a;
        }
        for (int j = 0; j < H; ++j) { Gh[j] = Ghprev[j] + ddh[j] + Ghp[j]; Ghp[j] = -ddh[j]; }
    }
    for (int j = 0; j < 2 * H; ++j) { dp[L->o_b_ih + j] += Gdm[j]; dp[L->o_b_hh + j] += Gdm[j]; }
}

/* ------------------------------------------------------------------------------------------ */
/* TCNN: tcnn.py:82-97.  6 -> C (1x1, bias) HS, 4 x depthwise k5 dil 1,2,4,8 (pad 2d) HS,       */
/* C -> 2 (1x1, no bias), + [I,Q] residual                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int C; int64_t o_w0, o_b0, o_dw[4], o_w5; } tcnn_layout_t;
static void tcnn_layout(const odpd_model_t* m, tcnn_layout_t* g) {
    int64_t C = m->hidden, o = 0;
    g->C = (int)C;
    g->o_w0 = o; o += 6 * C; g->o_b0 = o; o += C;
    for (int l = 0; l < 4; ++l) { g->o_dw[l] = o; o += 5 * C; }
    g->o_w5 = o;
}
/* act[l] (T,C) are PRE-activations of stage l = 0..4; returns y. work arrays supplied by caller */
static void tcnn_seq_fwd(const tcnn_layout_t* L, const real* p, int T, const real* x, real* y, real* feat, real* pre) {
    int C = L->C;
    for (int t = 0; t < T; ++t) {
        real* f = feat + t * 6;
        real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
        f[0] = I; f[1] = Q; f[2] = a; f[3] = a * a * a; f[4] = Q / a; f[5] = I / a;
        for (int c = 0; c < C; ++c) {
            real v = p[L->o_b0 + c];
            for (int i = 0; i < 6; ++i) v += p[L->o_w0 + c * 6 + i] * f[i];
            pre[(0 * T + t) * C + c] = v;
        }
    }
    for (int l = 0; l < 4; ++l) {
        int d = 1 << l;
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < C; ++c) {
                real v = 0;
                for (int k = 0; k < 5; ++k) {
                    int tt = t + d * (k - 2);
                    if (tt >= 0 && tt < T) v += p[L->o_dw[l] + c * 5 + k] * hswish(pre[(l * T + tt) * C + c]);
                }
                pre[((l + 1) * T + t) * C + c] = v;
            }
    }
    for (int t = 0; t < T; ++t)
        for (int o = 0; o < 2; ++o) {
            real v = 0;
            for (int c = 0; c < C; ++c) v += p[L->o_w5 + o * C + c] * hswish(pre[(4 * T + t) * C + c]);
            y[2 * t + o] = v + x[2 * t + o];
        }
}
static void tcnn_seq_bwd(const tcnn_layout_t* L, const real* p, int T, const real* x, const real* dy, const real* feat,
                         const real* pre, real* dp, real* dx, real* gcur, real* gnext) {
    int C = L->C;
    /* gcur = dL/d(hswish(pre[4])) */
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c) {
            real g = 0;
            for (int o = 0; o < 2; ++o) { g += dy[2 * t + o] * p[L->o_w5 + o * C + c]; dp[L->o_w5 + o * C + c] += dy[2 * t + o] * hswish(pre[(4 * T + t) * C + c]); }
            gcur[t * C + c] = g;
        }
    for (int l = 3; l >= 0; --l) {
        int d = 1 << l;
        memset(gnext, 0, sizeof(real) * T * C);
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < C; ++c) {
                real gp = gcur[t * C + c] * hswish_grad(pre[((l + 1) * T + t) * C + c]);
                for (int k = 0; k < 5; ++k) {
                    int tt = t + d * (k - 2);
                    if (tt >= 0 && tt < T) {
                        dp[L->o_dw[l] + c * 5 + k] += gp * hswish(pre[(l * T + tt) * C + c]);
                        gnext[tt * C + c] += gp * p[L->o_dw[l] + c * 5 + k];
                    }
                }
            }
        real* tsw = gcur; gcur = gnext; gnext = tsw;
    }
    for (int t = 0; t < T; ++t) {
        real df[6] = {0};
        for (int c = 0; c < C; ++c) {
            real gp = gcur[t * C + c] * hswish_grad(pre[(0 * T + t) * C + c]);
            dp[L->o_b0 + c] += gp;
            for (int i = 0; i < 6; ++i) { dp[L->o_w0 + c * 6 + i] += gp * feat[t * 6 + i]; df[i] += gp * p[L->o_w0 + c * 6 + i]; }
        }
        if (dx) {
            real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
            real da = df[2] + (real)3 * a * a * df[3] - (Q / a2) * df[4] - (I / a2) * df[5];
            dx[2 * t] = dy[2 * t] + df[0] + df[5] / a + da * I / a;
            dx[2 * t + 1] = dy[2 * t + 1] + df[1] + df[4] / a + da * Q / a;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* NeuralTX: neuraltx.py:116-137.  (The torch.fft.fft of :118 runs over a trailing dimension of size 1: the identity.)
 * Complex 5-tap FIR with zero padding 2 (:122-123: f_I = cI*xI - cQ*xQ, f_Q = cQ*xI + cI*xQ, Conv1d = cross-correlation);
 * features [f_I, f_Q, a, a^3], a = |f| (:124-129); the TCNN stack 4 -> C (1x1, bias) HS, 4 x depthwise k5 dil 1,2,4,8 HS,
 * C -> 2 (1x1) (:21-37); y = net + IQ_match f + f (:135).  Parameter order: conv_I.weight (5), conv_Q.weight (5),
 * network.0.weight (C,4), network.0.bias (C), network.{2,4,6,8}.weight (C,5), network.10.weight (2,C), IQ_match.weight (2,2). */
/* ------------------------------------------------------------------------------------------ */
/* `--quant` (bits_w > 0): the surgery's layer map holds nn.Conv2d and nn.Linear (quant_envs.py:145-148), so the Conv1d FIRs and stack stay
 * float and IQ_match becomes a bias-free INT_Linear (its three scales behind IQ_match.weight).  No module is named fc_out, so
 * set_last_layer_quant (:276-284) marks nothing: the 16-bit output quantiser never runs, train and eval mode compute the same. */
typedef struct { int C, q, bits_w, bits_a; int64_t o_ci, o_cq, o_w0, o_b0, o_dw[4], o_w5, o_m; } ntx_layout_t;
static void ntx_layout(const odpd_model_t* m, ntx_layout_t* g) {
    int64_t C = m->hidden, o = 0;
    g->C = (int)C;
    g->q = m->bits_w > 0; g->bits_w = m->bits_w; g->bits_a = m->bits_a;
    g->o_ci = o; o += 5; g->o_cq = o; o += 5;
    g->o_w0 = o; o += 4 * C; g->o_b0 = o; o += C;
    for (int l = 0; l < 4; ++l) { g->o_dw[l] = o; o += 5 * C; }
    g->o_w5 = o; o += 2 * C;
    g->o_m = o;
}
/* feat (T,4) = [f_I, f_Q, a, a^3]; pre (5,T,C) pre-activations of the five stages */
static void ntx_seq_fwd(const ntx_layout_t* L, const real* p, int T, const real* x, real* y, real* feat, real* pre) {
    const int C = L->C;
    for (int t = 0; t < T; ++t) {
        real fi = 0, fq = 0;
        for (int k = 0; k < 5; ++k) {
            const int tt = t + k - 2;
            if (tt < 0 || tt >= T) continue;
            fi += p[L->o_ci + k] * x[2 * tt] - p[L->o_cq + k] * x[2 * tt + 1];
            fq += p[L->o_cq + k] * x[2 * tt] + p[L->o_ci + k] * x[2 * tt + 1];
        }
        real* f = feat + t * 4;
        const real a = (real)sqrt((double)(fi * fi + fq * fq));
        f[0] = fi; f[1] = fq; f[2] = a; f[3] = a * a * a;
        for (int c = 0; c < C; ++c) {
            real v = p[L->o_b0 + c];
            for (int i = 0; i < 4; ++i) v += p[L->o_w0 + c * 4 + i] * f[i];
            pre[(0 * T + t) * C + c] = v;
        }
    }
    for (int l = 0; l < 4; ++l) {
        const int d = 1 << l;
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < C; ++c) {
                real v = 0;
                for (int k = 0; k < 5; ++k) {
                    const int tt = t + d * (k - 2);
                    if (tt >= 0 && tt < T) v += p[L->o_dw[l] + c * 5 + k] * hswish(pre[(l * T + tt) * C + c]);
                }
                pre[((l + 1) * T + t) * C + c] = v;
            }
    }
    for (int t = 0; t < T; ++t)
        for (int o = 0; o < 2; ++o) {
            real v = 0;
            for (int c = 0; c < C; ++c) v += p[L->o_w5 + o * C + c] * hswish(pre[(4 * T + t) * C + c]);
            const real* f = feat + t * 4;
            if (L->q) {      /* (out + IQ_match(q_a(f))) + f, the INT_Linear on exact grid sums */
                const real sw = q_pow2(p[L->o_m + 4]), sa = q_pow2(p[L->o_m + 5]);
                const double lin = (double)q_apply(p[L->o_m + 2 * o], sw, L->bits_w, NULL) * (double)q_apply(f[0], sa, L->bits_a, NULL) +
                                   (double)q_apply(p[L->o_m + 2 * o + 1], sw, L->bits_w, NULL) * (double)q_apply(f[1], sa, L->bits_a, NULL);
                y[2 * t + o] = (v + (real)lin) + f[o];
            } else
            y[2 * t + o] = v + p[L->o_m + 2 * o] * f[0] + p[L->o_m + 2 * o + 1] * f[1] + f[o];
        }
}
/* dfq: scratch (T,2) for dL/d(f_I, f_Q) */
static void ntx_seq_bwd(const ntx_layout_t* L, const real* p, int T, const real* x, const real* dy, const real* feat,
                        const real* pre, real* dp, real* dx, real* gcur, real* gnext, real* dfq) {
    const int C = L->C;
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c) {
            real g = 0;
            for (int o = 0; o < 2; ++o) { g += dy[2 * t + o] * p[L->o_w5 + o * C + c]; dp[L->o_w5 + o * C + c] += dy[2 * t + o] * hswish(pre[(4 * T + t) * C + c]); }
            gcur[t * C + c] = g;
        }
    for (int l = 3; l >= 0; --l) {
        const int d = 1 << l;
        memset(gnext, 0, sizeof(real) * T * C);
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < C; ++c) {
                const real gp = gcur[t * C + c] * hswish_grad(pre[((l + 1) * T + t) * C + c]);
                for (int k = 0; k < 5; ++k) {
                    const int tt = t + d * (k - 2);
                    if (tt >= 0 && tt < T) {
                        dp[L->o_dw[l] + c * 5 + k] += gp * hswish(pre[(l * T + tt) * C + c]);
                        gnext[tt * C + c] += gp * p[L->o_dw[l] + c * 5 + k];
                    }
                }
            }
        real* tsw = gcur; gcur = gnext; gnext = tsw;
    }
    for (int t = 0; t < T; ++t) {
        real df[4] = {0, 0, 0, 0};
        const real* f = feat + t * 4;
        for (int c = 0; c < C; ++c) {
            const real gp = gcur[t * C + c] * hswish_grad(pre[(0 * T + t) * C + c]);
            dp[L->o_b0 + c] += gp;
            for (int i = 0; i < 4; ++i) { dp[L->o_w0 + c * 4 + i] += gp * f[i]; df[i] += gp * p[L->o_w0 + c * 4 + i]; }
        }
        /* IQ_match and identity skip of f */
        if (L->q) {
            const real sw = q_pow2(p[L->o_m + 4]), sa = q_pow2(p[L->o_m + 5]);
            real pass[2], fq[2];
            for (int j = 0; j < 2; ++j) fq[j] = q_apply(f[j], sa, L->bits_a, &pass[j]);
            for (int o = 0; o < 2; ++o)
                for (int j = 0; j < 2; ++j) {
                    real mk, wq = q_apply(p[L->o_m + 2 * o + j], sw, L->bits_w, &mk);
                    dp[L->o_m + 2 * o + j] += dy[2 * t + o] * fq[j] * mk;
                    df[j] += pass[j] * dy[2 * t + o] * wq;
                }
        } else
        for (int o = 0; o < 2; ++o) {
            dp[L->o_m + 2 * o] += dy[2 * t + o] * f[0];
            dp[L->o_m + 2 * o + 1] += dy[2 * t + o] * f[1];
            df[0] += dy[2 * t + o] * p[L->o_m + 2 * o];
            df[1] += dy[2 * t + o] * p[L->o_m + 2 * o + 1];
        }
        df[0] += dy[2 * t]; df[1] += dy[2 * t + 1];
        const real a = f[2], da = df[2] + (real)3 * a * a * df[3];      /* a = |f|: da/df = f / a */
        dfq[2 * t] = df[0] + da * f[0] / a;
        dfq[2 * t + 1] = df[1] + da * f[1] / a;
    }
    if (dx) memset(dx, 0, sizeof(real) * 2 * T);
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < 5; ++k) {
            const int tt = t + k - 2;
            if (tt < 0 || tt >= T) continue;
            const real gi = dfq[2 * t], gq = dfq[2 * t + 1], xi = x[2 * tt], xq = x[2 * tt + 1];
            dp[L->o_ci + k] += gi * xi + gq * xq;
            dp[L->o_cq + k] += gq * xi - gi * xq;
            if (dx) {
                dx[2 * tt] += p[L->o_ci + k] * gi + p[L->o_cq + k] * gq;
                dx[2 * tt + 1] += p[L->o_ci + k] * gq - p[L->o_cq + k] * gi;
            }
        }
}

/* ------------------------------------------------------------------------------------------ */
/* PGJANET: pgjanet.py:26-76                                                                    */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int H; int64_t o_wa, o_ba, o_wp1, o_bp1, o_wp2, o_bp2, o_wf, o_bf, o_wg, o_bg, o_wo, o_bo; } pgj_layout_t;
static void pgj_layout(const odpd_model_t* m, pgj_layout_t* g) {
    int64_t H = m->hidden, o = 0;
    g->H = (int)H;
    g->o_wa = o; o += H * (H + 1); g->o_ba = o; o += H;
    g->o_wp1 = o; o += H * (H + 1); g->o_bp1 = o; o += H;
    g->o_wp2 = o; o += H * (H + 1); g->o_bp2 = o; o += H;
    g->o_wf = o; o += H * 2 * H; g->o_bf = o; o += H;
    g->o_wg = o; o += H * 2 * H; g->o_bg = o; o += H;
    g->o_wo = o; o += 2 * H; g->o_bo = o; o += 2;
}
typedef struct { real amp, ct, st, hp[MAXH], an[MAXH], p1[MAXH], p2[MAXH], u[MAXH], f[MAXH], g[MAXH], h[MAXH]; } pgj_step_t;
static void pgj_seq_fwd(const pgj_layout_t* L, const real* p, int T, const real* x, real* y, pgj_step_t* S) {
    int H = L->H;
    real h[MAXH] = {0};
    pgj_step_t tmp;
    for (int t = 0; t < T; ++t) {
        pgj_step_t* s = S ? &S[t] : &tmp;
        real I = x[2 * t], Q = x[2 * t + 1];
        s->amp = (real)sqrt((double)(I * I + Q * Q));
        real th = (real)atan2((double)Q, (double)I);
        s->ct = (real)cos((double)th); s->st = (real)sin((double)th);
        for (int j = 0; j < H; ++j) s->hp[j] = h[j];
        for (int j = 0; j < H; ++j) {
            real a = p[L->o_ba + j], b = p[L->o_bp1 + j], c = p[L->o_bp2 + j];
            for (int i = 0; i < H; ++i) {
                a += p[L->o_wa + j * (H + 1) + i] * h[i]; b += p[L->o_wp1 + j * (H + 1) + i] * h[i]; c += p[L->o_wp2 + j * (H + 1) + i] * h[i];
            }
            a += p[L->o_wa + j * (H + 1) + H] * s->amp; b += p[L->o_wp1 + j * (H + 1) + H] * s->ct; c += p[L->o_wp2 + j * (H + 1) + H] * s->st;
            s->an[j] = tanhr(a); s->p1[j] = tanhr(b); s->p2[j] = tanhr(c);
            s->u[j] = s->an[j] * s->p1[j] * s->p2[j] * ((real)1 - s->an[j]) * ((real)1 - s->p1[j]) * ((real)1 - s->p2[j]);
        }
        real hn[MAXH];
        for (int j = 0; j < H; ++j) {
            real a = p[L->o_bf + j], b = p[L->o_bg + j];
            for (int i = 0; i < H; ++i) {
                a += p[L->o_wf + j * 2 * H + i] * h[i] + p[L->o_wf + j * 2 * H + H + i] * s->u[i];
                b += p[L->o_wg + j * 2 * H + i] * h[i] + p[L->o_wg + j * 2 * H + H + i] * s->u[i];
            }
            s->f[j] = sigm(a); s->g[j] = tanhr(b);
            hn[j] = s->f[j] * h[j] + ((real)1 - s->f[j]) * s->g[j];
        }
        for (int j = 0; j < H; ++j) { h[j] = hn[j]; s->h[j] = hn[j]; }
        for (int c = 0; c < 2; ++c) {
            real a = p[L->o_bo + c];
            for (int j = 0; j < H; ++j) a += p[L->o_wo + c * H + j] * h[j];
            y[2 * t + c] = a;
        }
    }
}
static void pgj_seq_bwd(const pgj_layout_t* L, const real* p, int T, const real* x, const real* dy, const pgj_step_t* S,
                        real* dp, real* dx) {
    int H = L->H;
    real dh[MAXH] = {0};
    for (int t = T - 1; t >= 0; --t) {
        const pgj_step_t* s = &S[t];
        for (int c = 0; c < 2; ++c) {
            real d = dy[2 * t + c];
            dp[L->o_bo + c] += d;
            for (int j = 0; j < H; ++j) { dp[L->o_wo + c * H + j] += d * s->h[j]; dh[j] += d * p[L->o_wo + c * H + j]; }
        }
        real dhp[MAXH] = {0}, du[MAXH] = {0};
        for (int j = 0; j < H; ++j) {
            real df = dh[j] * (s->hp[j] - s->g[j]), dg = dh[j] * ((real)1 - s->f[j]);
            dhp[j] += dh[j] * s->f[j];
            real dfp = df * s->f[j] * ((real)1 - s->f[j]), dgp = dg * ((real)1 - s->g[j] * s->g[j]);
            dp[L->o_bf + j] += dfp; dp[L->o_bg + j] += dgp;
            for (int i = 0; i < H; ++i) {
                dp[L->o_wf + j * 2 * H + i] += dfp * s->hp[i]; dp[L->o_wf + j * 2 * H + H + i] += dfp * s->u[i];
                dp[L->o_wg + j * 2 * H + i] += dgp * s->hp[i]; dp[L->o_wg + j * 2 * H + H + i] += dgp * s->u[i];
                dhp[i] += dfp * p[L->o_wf + j * 2 * H + i] + dgp * p[L->o_wg + j * 2 * H + i];
                du[i] += dfp * p[L->o_wf + j * 2 * H + H + i] + dgp * p[L->o_wg + j * 2 * H + H + i];
            }
        }
        real damp = 0, dct = 0, dst = 0;
        for (int j = 0; j < H; ++j) {
            real a = s->an[j], b = s->p1[j], c = s->p2[j];
            /* u = A(a) A(b) A(c), A(v) = v (1 - v), A'(v) = 1 - 2 v */
            real Aa = a * ((real)1 - a), Ab = b * ((real)1 - b), Ac = c * ((real)1 - c);
            real da = du[j] * ((real)1 - (real)2 * a) * Ab * Ac, db = du[j] * Aa * ((real)1 - (real)2 * b) * Ac,
                 dc = du[j] * Aa * Ab * ((real)1 - (real)2 * c);
            real dap = da * ((real)1 - a * a), dbp = db * ((real)1 - b * b), dcp = dc * ((real)1 - c * c);
            dp[L->o_ba + j] += dap; dp[L->o_bp1 + j] += dbp; dp[L->o_bp2 + j] += dcp;
            for (int i = 0; i < H; ++i) {
                dp[L->o_wa + j * (H + 1) + i] += dap * s->hp[i]; dp[L->o_wp1 + j * (H + 1) + i] += dbp * s->hp[i]; dp[L->o_wp2 + j * (H + 1) + i] += dcp * s->hp[i];
                dhp[i] += dap * p[L->o_wa + j * (H + 1) + i] + dbp * p[L->o_wp1 + j * (H + 1) + i] + dcp * p[L->o_wp2 + j * (H + 1) + i];
            }
            dp[L->o_wa + j * (H + 1) + H] += dap * s->amp; dp[L->o_wp1 + j * (H + 1) + H] += dbp * s->ct; dp[L->o_wp2 + j * (H + 1) + H] += dcp * s->st;
            damp += dap * p[L->o_wa + j * (H + 1) + H]; dct += dbp * p[L->o_wp1 + j * (H + 1) + H]; dst += dcp * p[L->o_wp2 + j * (H + 1) + H];
        }
        for (int j = 0; j < H; ++j) dh[j] = dhp[j];
        if (dx) {
            /* theta = atan2(Q,I): dtheta = -sin dct + cos dst; dtheta/dI = -Q/a^2, dtheta/dQ = I/a^2 */
            real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q;
            real dth = -s->st * dct + s->ct * dst;
            dx[2 * t] = damp * I / s->amp - dth * Q / a2;
            dx[2 * t + 1] = damp * Q / s->amp + dth * I / a2;
        }
    }
}

/* `--quant` on pgjanet (bits_w > 0): its six nn.Linear (W_a, W_p1, W_p2 on [h, scalar]; W_f, W_g on [h, u]; W_o on h) become INT_Linear
 * (qlin_fwd / qlin_bwd: each quantises its input on an activation grid of its own), three scales behind each layer's bias; the tanh / sigmoid
 * calls are functional and stay float; no module is named fc_out, so no output quantiser runs in either mode (quant_envs.py:276-284). */
typedef struct { int H, bw, ba; int64_t ow[6], ob[6], oq[6], P; } pgq_layout_t;      /* 0 W_a 1 W_p1 2 W_p2 3 W_f 4 W_g 5 W_o */
static void pgq_layout(const odpd_model_t* m, pgq_layout_t* g) {
    int64_t H = m->hidden, o = 0;
    g->H = (int)H; g->bw = m->bits_w; g->ba = m->bits_a;
    for (int l = 0; l < 6; ++l) {
        const int64_t nin = l < 3 ? H + 1 : (l < 5 ? 2 * H : H), nout = l < 5 ? H : 2;
        g->ow[l] = o; o += nout * nin; g->ob[l] = o; o += nout; g->oq[l] = o; o += 3;
    }
    g->P = o;
}
typedef struct {
    real amp, ct, st, hp[MAXH], an[MAXH], p1[MAXH], p2[MAXH], f[MAXH], g[MAXH];
    real inq[6][2 * MAXH], pass[6][2 * MAXH];      /* q_a(input), pass mask of each INT_Linear */
} pgq_step_t;
static void pgq_seq_fwd(const pgq_layout_t* L, const real* p, int T, const real* x, real* y, pgq_step_t* S) {
    const int H = L->H;
    real h[MAXH] = {0};
    pgq_step_t tmp;
    for (int t = 0; t < T; ++t) {
        pgq_step_t* s = S ? &S[t] : &tmp;
        const real I = x[2 * t], Q = x[2 * t + 1];
        s->amp = (real)sqrt((double)(I * I + Q * Q));
        const real th = (real)atan2((double)Q, (double)I);
        s->ct = (real)cos((double)th); s->st = (real)sin((double)th);
        real in[2 * MAXH], pre[3][MAXH], u[MAXH];
        for (int j = 0; j < H; ++j) { s->hp[j] = h[j]; in[j] = h[j]; }
        const real sc[3] = {s->amp, s->ct, s->st};
        for (int l = 0; l < 3; ++l) {
            in[H] = sc[l];
            qlin_fwd(p, L->ow[l], L->ob[l], L->oq[l], H, H + 1, L->bw, L->ba, in, s->inq[l], s->pass[l], pre[l]);
        }
        for (int j = 0; j < H; ++j) {
            s->an[j] = tanhr(pre[0][j]); s->p1[j] = tanhr(pre[1][j]); s->p2[j] = tanhr(pre[2][j]);
            u[j] = s->an[j] * s->p1[j] * s->p2[j] * ((real)1 - s->an[j]) * ((real)1 - s->p1[j]) * ((real)1 - s->p2[j]);
            in[H + j] = u[j];
        }
        qlin_fwd(p, L->ow[3], L->ob[3], L->oq[3], H, 2 * H, L->bw, L->ba, in, s->inq[3], s->pass[3], pre[0]);
        qlin_fwd(p, L->ow[4], L->ob[4], L->oq[4], H, 2 * H, L->bw, L->ba, in, s->inq[4], s->pass[4], pre[1]);
        for (int j = 0; j < H; ++j) {
            s->f[j] = sigm(pre[0][j]); s->g[j] = tanhr(pre[1][j]);
            h[j] = s->f[j] * h[j] + ((real)1 - s->f[j]) * s->g[j];
        }
        qlin_fwd(p, L->ow[5], L->ob[5], L->oq[5], 2, H, L->bw, L->ba, h, s->inq[5], s->pass[5], &y[2 * t]);
    }
}
static void pgq_seq_bwd(const pgq_layout_t* L, const real* p, int T, const real* x, const real* dy, const pgq_step_t* S, real* dp, real* dx) {
    const int H = L->H;
    real dh[MAXH] = {0};
    for (int t = T - 1; t >= 0; --t) {
        const pgq_step_t* s = &S[t];
        qlin_bwd(p, L->ow[5], L->ob[5], L->oq[5], 2, H, L->bw, &dy[2 * t], s->inq[5], s->pass[5], dp, dh);
        real dhp[MAXH] = {0}, dfp[MAXH], dgp[MAXH], din[2 * MAXH] = {0};
        for (int j = 0; j < H; ++j) {
            const real df = dh[j] * (s->hp[j] - s->g[j]), dg = dh[j] * ((real)1 - s->f[j]);
            dhp[j] += dh[j] * s->f[j];
            dfp[j] = df * s->f[j] * ((real)1 - s->f[j]); dgp[j] = dg * ((real)1 - s->g[j] * s->g[j]);
        }
        qlin_bwd(p, L->ow[3], L->ob[3], L->oq[3], H, 2 * H, L->bw, dfp, s->inq[3], s->pass[3], dp, din);
        qlin_bwd(p, L->ow[4], L->ob[4], L->oq[4], H, 2 * H, L->bw, dgp, s->inq[4], s->pass[4], dp, din);
        real dpre[3][MAXH], dsc[3];
        for (int j = 0; j < H; ++j) {
            dhp[j] += din[j];
            const real du = din[H + j], a = s->an[j], b = s->p1[j], c = s->p2[j];
            const real Aa = a * ((real)1 - a), Ab = b * ((real)1 - b), Ac = c * ((real)1 - c);
            dpre[0][j] = du * ((real)1 - (real)2 * a) * Ab * Ac * ((real)1 - a * a);
            dpre[1][j] = du * Aa * ((real)1 - (real)2 * b) * Ac * ((real)1 - b * b);
            dpre[2][j] = du * Aa * Ab * ((real)1 - (real)2 * c) * ((real)1 - c * c);
        }
        for (int l = 0; l < 3; ++l) {
            real d1[MAXH + 1] = {0};
            qlin_bwd(p, L->ow[l], L->ob[l], L->oq[l], H, H + 1, L->bw, dpre[l], s->inq[l], s->pass[l], dp, d1);
            for (int j = 0; j < H; ++j) dhp[j] += d1[j];
            dsc[l] = d1[H];
        }
        for (int j = 0; j < H; ++j) dh[j] = dhp[j];
        if (dx) {
            const real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q;
            const real dth = -s->st * dsc[1] + s->ct * dsc[2];
            dx[2 * t] = dsc[0] * I / s->amp - dth * Q / a2;
            dx[2 * t + 1] = dsc[0] * Q / s->amp + dth * I / a2;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* DVRJANET: dvrjanet.py:44-101.  Two states h_I, h_Q (both start at h_0 = 0); per step, with hs = h_I + h_Q, mag = |x|,
 * theta = atan2(Q, I) (:58-63):  th = W_ptheta theta + W_ph hs (:66);  a = DVR(W_ax mag + W_ah hs) with DVR(v) = sum_k c_k |v - k/K|,
 * k = 1..K (:32-42, :69-72);  f = sigmoid(W_f hs + b) (:80);  g_c = tanh(W_ccos [h_I, a cos th] + b), g_s = tanh(W_csin [h_Q, a sin th]
 * + b) (:83-86);  h_I = f h_I + (1-f) g_c, h_Q = f h_Q + (1-f) g_s (:89-90);  y = (W_o1 h_I + b, W_o2 h_Q + b) (:93-94).
 * K = num_dvr_units rides in odpd_model_t::bits_w.  Parameter order (named_parameters: the module's own `cs` first): cs (K),
 * W_ph (H,H), W_ptheta (H,1), W_ah (H,H), W_ax (H,1), W_f (H,H)+b, W_ccos (H,2H)+b, W_csin (H,2H)+b, W_o1 (1,H)+b, W_o2 (1,H)+b. */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int H, K; int64_t o_cs, o_wph, o_wpt, o_wah, o_wax, o_wf, o_bf, o_wcc, o_bcc, o_wcs, o_bcs, o_wo1, o_bo1, o_wo2, o_bo2, P; } dvr_layout_t;
static void dvr_layout(const odpd_model_t* m, dvr_layout_t* g) {
    int64_t H = m->hidden, K = m->bits_w, o = 0;
    g->H = (int)H; g->K = (int)K;
    g->o_cs = o; o += K;
    g->o_wph = o; o += H * H; g->o_wpt = o; o += H; g->o_wah = o; o += H * H; g->o_wax = o; o += H;
    g->o_wf = o; o += H * H; g->o_bf = o; o += H;
    g->o_wcc = o; o += 2 * H * H; g->o_bcc = o; o += H;
    g->o_wcs = o; o += 2 * H * H; g->o_bcs = o; o += H;
    g->o_wo1 = o; o += H; g->o_bo1 = o; o += 1; g->o_wo2 = o; o += H; g->o_bo2 = o; o += 1;
    g->P = o;
}
typedef struct { real mag, theta, hIp[MAXH], hQp[MAXH], th[MAXH], ap[MAXH], at[MAXH], f[MAXH], gc[MAXH], gs[MAXH], hI[MAXH], hQ[MAXH]; } dvr_step_t;
static void dvr_seq_fwd(const dvr_layout_t* L, const real* p, int T, const real* x, real* y, dvr_step_t* S) {
    const int H = L->H, K = L->K;
    real hI[MAXH] = {0}, hQ[MAXH] = {0};
    dvr_step_t tmp;
    for (int t = 0; t < T; ++t) {
        dvr_step_t* s = S ? &S[t] : &tmp;
        const real I = x[2 * t], Q = x[2 * t + 1];
        s->mag = (real)sqrt((double)(I * I + Q * Q));
        s->theta = (real)atan2((double)Q, (double)I);
        real hs[MAXH];
        for (int j = 0; j < H; ++j) { s->hIp[j] = hI[j]; s->hQp[j] = hQ[j]; hs[j] = hI[j] + hQ[j]; }
        for (int j = 0; j < H; ++j) {
            real a0 = 0, a1 = 0, a2 = p[L->o_bf + j];
            for (int i = 0; i < H; ++i) { a0 += p[L->o_wph + j * H + i] * hs[i]; a1 += p[L->o_wah + j * H + i] * hs[i]; a2 += p[L->o_wf + j * H + i] * hs[i]; }
            s->th[j] = p[L->o_wpt + j] * s->theta + a0;
            s->ap[j] = p[L->o_wax + j] * s->mag + a1;
            real at = 0;
            for (int k = 1; k <= K; ++k) at += (real)fabs((double)(s->ap[j] - (real)k / (real)K)) * p[L->o_cs + k - 1];
            s->at[j] = at;
            s->f[j] = sigm(a2);
        }
        real vc[MAXH], vs[MAXH];
        for (int j = 0; j < H; ++j) { vc[j] = s->at[j] * (real)cos((double)s->th[j]); vs[j] = s->at[j] * (real)sin((double)s->th[j]); }
        for (int j = 0; j < H; ++j) {
            real c = p[L->o_bcc + j], q = p[L->o_bcs + j];
            for (int i = 0; i < H; ++i) {
                c += p[L->o_wcc + j * 2 * H + i] * hI[i] + p[L->o_wcc + j * 2 * H + H + i] * vc[i];
                q += p[L->o_wcs + j * 2 * H + i] * hQ[i] + p[L->o_wcs + j * 2 * H + H + i] * vs[i];
            }
            s->gc[j] = tanhr(c); s->gs[j] = tanhr(q);
        }
        real y0 = p[L->o_bo1], y1 = p[L->o_bo2];
        for (int j = 0; j < H; ++j) {
            hI[j] = s->f[j] * hI[j] + ((real)1 - s->f[j]) * s->gc[j];
            hQ[j] = s->f[j] * hQ[j] + ((real)1 - s->f[j]) * s->gs[j];
            s->hI[j] = hI[j]; s->hQ[j] = hQ[j];
            y0 += p[L->o_wo1 + j] * hI[j]; y1 += p[L->o_wo2 + j] * hQ[j];
        }
        y[2 * t] = y0; y[2 * t + 1] = y1;
    }
}
static void dvr_seq_bwd(const dvr_layout_t* L, const real* p, int T, const real* x, const real* dy, const dvr_step_t* S, real* dp, real* dx) {
    const int H = L->H, K = L->K;
    real dhI[MAXH] = {0}, dhQ[MAXH] = {0};
    for (int t = T - 1; t >= 0; --t) {
        const dvr_step_t* s = &S[t];
        const real d0 = dy[2 * t], d1 = dy[2 * t + 1];
        dp[L->o_bo1] += d0; dp[L->o_bo2] += d1;
        real dfp[MAXH], dgc[MAXH], dgs[MAXH], nI[MAXH], nQ[MAXH], hs[MAXH], vc[MAXH], vs[MAXH], co[MAXH], si[MAXH];
        for (int j = 0; j < H; ++j) {
            dp[L->o_wo1 + j] += d0 * s->hI[j]; dp[L->o_wo2 + j] += d1 * s->hQ[j];
            const real gI = dhI[j] + d0 * p[L->o_wo1 + j], gQ = dhQ[j] + d1 * p[L->o_wo2 + j];
            const real df = gI * (s->hIp[j] - s->gc[j]) + gQ * (s->hQp[j] - s->gs[j]);
            dfp[j] = df * s->f[j] * ((real)1 - s->f[j]);
            dgc[j] = gI * ((real)1 - s->f[j]) * ((real)1 - s->gc[j] * s->gc[j]);
            dgs[j] = gQ * ((real)1 - s->f[j]) * ((real)1 - s->gs[j] * s->gs[j]);
            nI[j] = gI * s->f[j]; nQ[j] = gQ * s->f[j];
            hs[j] = s->hIp[j] + s->hQp[j];
            co[j] = (real)cos((double)s->th[j]); si[j] = (real)sin((double)s->th[j]);
            vc[j] = s->at[j] * co[j]; vs[j] = s->at[j] * si[j];
            dp[L->o_bf + j] += dfp[j]; dp[L->o_bcc + j] += dgc[j]; dp[L->o_bcs + j] += dgs[j];
        }
        real dvc[MAXH] = {0}, dvs[MAXH] = {0};
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < H; ++i) {
                dp[L->o_wcc + j * 2 * H + i] += dgc[j] * s->hIp[i]; dp[L->o_wcc + j * 2 * H + H + i] += dgc[j] * vc[i];
                dp[L->o_wcs + j * 2 * H + i] += dgs[j] * s->hQp[i]; dp[L->o_wcs + j * 2 * H + H + i] += dgs[j] * vs[i];
                nI[i] += p[L->o_wcc + j * 2 * H + i] * dgc[j]; dvc[i] += p[L->o_wcc + j * 2 * H + H + i] * dgc[j];
                nQ[i] += p[L->o_wcs + j * 2 * H + i] * dgs[j]; dvs[i] += p[L->o_wcs + j * 2 * H + H + i] * dgs[j];
            }
        real dth[MAXH], dap[MAXH], gth = 0, gmag = 0;
        for (int j = 0; j < H; ++j) {
            const real dat = dvc[j] * co[j] + dvs[j] * si[j];
            dth[j] = s->at[j] * (dvs[j] * co[j] - dvc[j] * si[j]);
            real slope = 0;
            for (int k = 1; k <= K; ++k) {
                const real u = s->ap[j] - (real)k / (real)K;
                dp[L->o_cs + k - 1] += dat * (real)fabs((double)u);
                slope += p[L->o_cs + k - 1] * (u > 0 ? (real)1 : (u < 0 ? (real)-1 : (real)0));     /* torch.abs: gradient 0 at 0 */
            }
            dap[j] = dat * slope;
            dp[L->o_wpt + j] += dth[j] * s->theta; dp[L->o_wax + j] += dap[j] * s->mag;
            gth += p[L->o_wpt + j] * dth[j]; gmag += p[L->o_wax + j] * dap[j];
        }
        for (int j = 0; j < H; ++j)
            for (int i = 0; i < H; ++i) {
                dp[L->o_wph + j * H + i] += dth[j] * hs[i]; dp[L->o_wah + j * H + i] += dap[j] * hs[i]; dp[L->o_wf + j * H + i] += dfp[j] * hs[i];
                const real g = p[L->o_wph + j * H + i] * dth[j] + p[L->o_wah + j * H + i] * dap[j] + p[L->o_wf + j * H + i] * dfp[j];
                nI[i] += g; nQ[i] += g;
            }
        for (int j = 0; j < H; ++j) { dhI[j] = nI[j]; dhQ[j] = nQ[j]; }
        if (dx) {   /* mag = |x|: d/dI = I/mag;  theta = atan2(Q, I): d/dI = -Q/mag^2, d/dQ = I/mag^2 */
            const real I = x[2 * t], Q = x[2 * t + 1], m2 = s->mag * s->mag;
            dx[2 * t] = gmag * I / s->mag - gth * Q / m2;
            dx[2 * t + 1] = gmag * Q / s->mag + gth * I / m2;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* BOJANET: bojanet.py:5-138.  M = 16-tap complex FIR bank of P = 6 filters over the zero-left-padded frame (:72-84):
 *   fi[p] = sum_m bI[p][m] I[t-15+m] - bQ[p][m] Q[t-15+m],  fq[p] = sum_m bQ[p][m] I[t-15+m] + bI[p][m] Q[t-15+m];
 * vector demodulator (:30-39): mag = sqrt(fi^2 + fq^2) + 1e-8, mag2 = mag^2, sin = fq / mag, cos = fi / mag;
 * envelope e = [mag(6), mag2(6)] (:86-87);  JANET cell (:91-94): f = s(W_fi e + b + W_fh h), g = tanh(W_gi e + b + W_gh h),
 * h = f h + (1 - f) g;  phase re-rotation (:41-53): unit j takes the phase of filter j mod 6 (hidden <= 18: the reference's
 * concatenation cannot build more);  read-outs (:103-104): A = w_I . (h cos) + b_I, Bq = w_Q . (h sin) + b_Q, y = (A - Bq, Bq + A).
 * Parameter order (named_parameters): fir_I (6,16), fir_Q (6,16), W_fi (H,12)+b, W_fh (H,H), W_gi (H,12)+b, W_gh (H,H),
 * W_out_I (1,H)+b, W_out_Q (1,H)+b. */
#define BOJ_P 6
#define BOJ_M 16
typedef struct { int H; int64_t o_bi, o_bq, o_wfi, o_bfi, o_wfh, o_wgi, o_bgi, o_wgh, o_woi, o_boi, o_woq, o_boq, P; } boj_layout_t;
static void boj_layout(const odpd_model_t* m, boj_layout_t* g) {
    int64_t H = m->hidden, o = 0;
    g->H = (int)H;
    g->o_bi = o; o += BOJ_P * BOJ_M; g->o_bq = o; o += BOJ_P * BOJ_M;
    g->o_wfi = o; o += H * 2 * BOJ_P; g->o_bfi = o; o += H; g->o_wfh = o; o += H * H;
    g->o_wgi = o; o += H * 2 * BOJ_P; g->o_bgi = o; o += H; g->o_wgh = o; o += H * H;
    g->o_woi = o; o += H; g->o_boi = o; o += 1; g->o_woq = o; o += H; g->o_boq = o; o += 1;
    g->P = o;
}
typedef struct { real fi[BOJ_P], fq[BOJ_P], m0[BOJ_P], mag[BOJ_P], hp[MAXH], f[MAXH], g[MAXH], h[MAXH]; } boj_step_t;
static void boj_seq_fwd(const boj_layout_t* L, const real* p, int T, const real* x, real* y, boj_step_t* S) {
    const int H = L->H;
    real h[MAXH] = {0};
    boj_step_t tmp;
    for (int t = 0; t < T; ++t) {
        boj_step_t* s = S ? &S[t] : &tmp;
        real e[2 * BOJ_P], co[BOJ_P], si[BOJ_P];
        for (int q = 0; q < BOJ_P; ++q) {
            real fi = 0, fq = 0;
            for (int m = 0; m < BOJ_M; ++m) {
                const int tt = t - (BOJ_M - 1) + m;
                if (tt < 0) continue;
                const real I = x[2 * tt], Q = x[2 * tt + 1], bi = p[L->o_bi + q * BOJ_M + m], bq = p[L->o_bq + q * BOJ_M + m];
                fi += bi * I - bq * Q; fq += bq * I + bi * Q;
            }
            s->fi[q] = fi; s->fq[q] = fq;
            s->m0[q] = (real)sqrt((double)(fi * fi + fq * fq));
            s->mag[q] = s->m0[q] + (real)1e-8;
            e[q] = s->mag[q]; e[BOJ_P + q] = s->mag[q] * s->mag[q];
            si[q] = fq / s->mag[q]; co[q] = fi / s->mag[q];
        }
        real y0 = p[L->o_boi], y1 = p[L->o_boq];
        for (int j = 0; j < H; ++j) {
            real pf = p[L->o_bfi + j], pg = p[L->o_bgi + j];
            for (int k = 0; k < 2 * BOJ_P; ++k) { pf += p[L->o_wfi + j * 2 * BOJ_P + k] * e[k]; pg += p[L->o_wgi + j * 2 * BOJ_P + k] * e[k]; }
            for (int k = 0; k < H; ++k) { pf += p[L->o_wfh + j * H + k] * h[k]; pg += p[L->o_wgh + j * H + k] * h[k]; }
            s->f[j] = sigm(pf); s->g[j] = tanhr(pg);
        }
        for (int j = 0; j < H; ++j) {
            s->hp[j] = h[j];
            h[j] = s->f[j] * h[j] + ((real)1 - s->f[j]) * s->g[j];
            s->h[j] = h[j];
            y0 += p[L->o_woi + j] * (h[j] * co[j % BOJ_P]); y1 += p[L->o_woq + j] * (h[j] * si[j % BOJ_P]);
        }
        y[2 * t] = y0 - y1; y[2 * t + 1] = y1 + y0;
    }
}
static void boj_seq_bwd(const boj_layout_t* L, const real* p, int T, const real* x, const real* dy, const boj_step_t* S, real* dp, real* dx) {
    const int H = L->H;
    real dh[MAXH] = {0};
    if (dx) for (int t = 0; t < 2 * T; ++t) dx[t] = 0;
    for (int t = T - 1; t >= 0; --t) {
        const boj_step_t* s = &S[t];
        const real dA = dy[2 * t] + dy[2 * t + 1], dB = dy[2 * t + 1] - dy[2 * t];
        dp[L->o_boi] += dA; dp[L->o_boq] += dB;
        real dco[BOJ_P] = {0}, dsi[BOJ_P] = {0}, de[2 * BOJ_P] = {0}, dfp[MAXH], dgp[MAXH], nh[MAXH];
        for (int j = 0; j < H; ++j) {
            const int q = j % BOJ_P;
            const real co = s->fi[q] / s->mag[q], si = s->fq[q] / s->mag[q];
            dp[L->o_woi + j] += dA * (s->h[j] * co); dp[L->o_woq + j] += dB * (s->h[j] * si);
            const real gh = dh[j] + dA * p[L->o_woi + j] * co + dB * p[L->o_woq + j] * si;
            dco[q] += dA * p[L->o_woi + j] * s->h[j]; dsi[q] += dB * p[L->o_woq + j] * s->h[j];
            dfp[j] = gh * (s->hp[j] - s->g[j]) * s->f[j] * ((real)1 - s->f[j]);
            dgp[j] = gh * ((real)1 - s->f[j]) * ((real)1 - s->g[j] * s->g[j]);
            nh[j] = gh * s->f[j];
            dp[L->o_bfi + j] += dfp[j]; dp[L->o_bgi + j] += dgp[j];
        }
        real e[2 * BOJ_P];
        for (int q = 0; q < BOJ_P; ++q) { e[q] = s->mag[q]; e[BOJ_P + q] = s->mag[q] * s->mag[q]; }
        for (int j = 0; j < H; ++j) {
            for (int k = 0; k < 2 * BOJ_P; ++k) {
                dp[L->o_wfi + j * 2 * BOJ_P + k] += dfp[j] * e[k]; dp[L->o_wgi + j * 2 * BOJ_P + k] += dgp[j] * e[k];
                de[k] += p[L->o_wfi + j * 2 * BOJ_P + k] * dfp[j] + p[L->o_wgi + j * 2 * BOJ_P + k] * dgp[j];
            }
            for (int k = 0; k < H; ++k) {
                dp[L->o_wfh + j * H + k] += dfp[j] * s->hp[k]; dp[L->o_wgh + j * H + k] += dgp[j] * s->hp[k];
                nh[k] += p[L->o_wfh + j * H + k] * dfp[j] + p[L->o_wgh + j * H + k] * dgp[j];
            }
        }
        for (int j = 0; j < H; ++j) dh[j] = nh[j];
        for (int q = 0; q < BOJ_P; ++q) {
            const real mag = s->mag[q];
            const real dmag = de[q] + (real)2 * mag * de[BOJ_P + q] - (dsi[q] * s->fq[q] + dco[q] * s->fi[q]) / (mag * mag);
            const real dfi = dco[q] / mag + dmag * s->fi[q] / s->m0[q], dfq = dsi[q] / mag + dmag * s->fq[q] / s->m0[q];
            for (int m = 0; m < BOJ_M; ++m) {
                const int tt = t - (BOJ_M - 1) + m;
                if (tt < 0) continue;
                const real I = x[2 * tt], Q = x[2 * tt + 1], bi = p[L->o_bi + q * BOJ_M + m], bq = p[L->o_bq + q * BOJ_M + m];
                dp[L->o_bi + q * BOJ_M + m] += dfi * I + dfq * Q; dp[L->o_bq + q * BOJ_M + m] += dfq * I - dfi * Q;
                if (dx) { dx[2 * tt] += dfi * bi + dfq * bq; dx[2 * tt + 1] += dfq * bi - dfi * bq; }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* APNRRU: apnrru.py:5-152.  16-tap complex FIR bank of 3 filters over the zero-left-padded frame (:66-90) plus the raw sample as a
 * fourth complex value; all four are rotated by r = conj(x_t) / |x_t| (:77-99) -> 8 features [re, im] x 4.  Complex state h (H) and
 * envelope state h_A (3), all 0 at the start.  Per step (:101-128): h <- h r (into the normalised frame); s = [h_I, h_Q, h_A] (n = 2H+3);
 * v = tanh(W_h tanh(W_u [feat, s] + b_u) + b_h);  s' = sigmoid(C s) + Z v  (RRU, :23-33; C scalar, Z (1,n));  h <- conj(r) s'_h (back);
 * y = (A - Bq, Bq + A) with A = w_I . h_I, Bq = w_Q . h_Q (no biases, :124-125).
 * Parameter order (named_parameters): fir_I (3,16), fir_Q (3,16), rru.C (1), rru.Z (1,n), rru.W_u (16, 8+n) + b (16),
 * rru.W_h (n,16) + b (n), output_layer_I (1,H), output_layer_Q (1,H). */
#define APN_F 3
#define APN_M 16
#define APN_NODE 16
#define APN_MAXN (2 * MAXH + 3)
typedef struct { int H, n; int64_t o_bi, o_bq, o_c, o_z, o_wu, o_bu, o_wh, o_bh, o_woi, o_woq, P; } apn_layout_t;
static void apn_layout(const odpd_model_t* m, apn_layout_t* g) {
    int64_t H = m->hidden, n = 2 * H + 3, o = 0;
    g->H = (int)H; g->n = (int)n;
    g->o_bi = o; o += APN_F * APN_M; g->o_bq = o; o += APN_F * APN_M;
    g->o_c = o; o += 1; g->o_z = o; o += n;
    g->o_wu = o; o += APN_NODE * (8 + n); g->o_bu = o; o += APN_NODE;
    g->o_wh = o; o += n * APN_NODE; g->o_bh = o; o += n;
    g->o_woi = o; o += H; g->o_woq = o; o += H;
    g->P = o;
}
typedef struct { real fi[4], fq[4], rr, ri, mag, feat[8], sp[APN_MAXN], v1[APN_NODE], v[APN_MAXN], sn[APN_MAXN], hI0[MAXH], hQ0[MAXH], hI[MAXH], hQ[MAXH]; } apn_step_t;
static void apn_seq_fwd(const apn_layout_t* L, const real* p, int T, const real* x, real* y, apn_step_t* S) {
    const int H = L->H, n = L->n;
    real hI[MAXH] = {0}, hQ[MAXH] = {0}, hA[3] = {0};
    apn_step_t tmp;
    for (int t = 0; t < T; ++t) {
        apn_step_t* s = S ? &S[t] : &tmp;
        const real I = x[2 * t], Q = x[2 * t + 1];
        s->mag = (real)sqrt((double)(I * I + Q * Q));
        s->rr = I / s->mag; s->ri = -Q / s->mag;
        for (int q = 0; q < APN_F; ++q) {
            real fi = 0, fq = 0;
            for (int m = 0; m < APN_M; ++m) {
                const int tt = t - (APN_M - 1) + m;
                if (tt < 0) continue;
                const real xi = x[2 * tt], xq = x[2 * tt + 1], bi = p[L->o_bi + q * APN_M + m], bq = p[L->o_bq + q * APN_M + m];
                fi += bi * xi - bq * xq; fq += bq * xi + bi * xq;
            }
            s->fi[q] = fi; s->fq[q] = fq;
        }
        s->fi[3] = I; s->fq[3] = Q;
        for (int k = 0; k < 4; ++k) {
            s->feat[2 * k] = s->rr * s->fi[k] - s->ri * s->fq[k];
            s->feat[2 * k + 1] = s->ri * s->fi[k] + s->rr * s->fq[k];
        }
        for (int j = 0; j < H; ++j) {
            s->hI0[j] = hI[j]; s->hQ0[j] = hQ[j];
            s->sp[j] = hI[j] * s->rr - hQ[j] * s->ri;
            s->sp[H + j] = hI[j] * s->ri + hQ[j] * s->rr;
        }
        for (int j = 0; j < 3; ++j) s->sp[2 * H + j] = hA[j];
        for (int o = 0; o < APN_NODE; ++o) {
            real a = p[L->o_bu + o];
            for (int k = 0; k < 8; ++k) a += p[L->o_wu + o * (8 + n) + k] * s->feat[k];
            for (int k = 0; k < n; ++k) a += p[L->o_wu + o * (8 + n) + 8 + k] * s->sp[k];
            s->v1[o] = tanhr(a);
        }
        for (int j = 0; j < n; ++j) {
            real a = p[L->o_bh + j];
            for (int o = 0; o < APN_NODE; ++o) a += p[L->o_wh + j * APN_NODE + o] * s->v1[o];
            s->v[j] = tanhr(a);
            s->sn[j] = sigm(p[L->o_c] * s->sp[j]) + p[L->o_z + j] * s->v[j];
        }
        real y0 = 0, y1 = 0;
        for (int j = 0; j < H; ++j) {
            hI[j] = s->sn[j] * s->rr + s->sn[H + j] * s->ri;
            hQ[j] = s->sn[H + j] * s->rr - s->sn[j] * s->ri;
            s->hI[j] = hI[j]; s->hQ[j] = hQ[j];
            y0 += p[L->o_woi + j] * hI[j]; y1 += p[L->o_woq + j] * hQ[j];
        }
        for (int j = 0; j < 3; ++j) hA[j] = s->sn[2 * H + j];
        y[2 * t] = y0 - y1; y[2 * t + 1] = y1 + y0;
    }
}
static void apn_seq_bwd(const apn_layout_t* L, const real* p, int T, const real* x, const real* dy, const apn_step_t* S, real* dp, real* dx) {
    const int H = L->H, n = L->n;
    real dhI[MAXH] = {0}, dhQ[MAXH] = {0}, dhA[3] = {0};
    if (dx) for (int t = 0; t < 2 * T; ++t) dx[t] = 0;
    for (int t = T - 1; t >= 0; --t) {
        const apn_step_t* s = &S[t];
        const real dA = dy[2 * t] + dy[2 * t + 1], dB = dy[2 * t + 1] - dy[2 * t], C = p[L->o_c];
        real dsn[APN_MAXN], dsp[APN_MAXN], dpre2[APN_MAXN], dv1[APN_NODE] = {0}, dpre1[APN_NODE], dfeat[8] = {0}, drr = 0, dri = 0;
        for (int j = 0; j < H; ++j) {
            dp[L->o_woi + j] += dA * s->hI[j]; dp[L->o_woq + j] += dB * s->hQ[j];
            const real gI = dhI[j] + dA * p[L->o_woi + j], gQ = dhQ[j] + dB * p[L->o_woq + j];
            /* h_I = sn_I rr + sn_Q ri,  h_Q = sn_Q rr - sn_I ri */
            dsn[j] = gI * s->rr - gQ * s->ri; dsn[H + j] = gI * s->ri + gQ * s->rr;
            drr += gI * s->sn[j] + gQ * s->sn[H + j]; dri += gI * s->sn[H + j] - gQ * s->sn[j];
        }
        for (int j = 0; j < 3; ++j) dsn[2 * H + j] = dhA[j];
        for (int j = 0; j < n; ++j) {
            const real sg = sigm(C * s->sp[j]), dsg = sg * ((real)1 - sg);
            dp[L->o_z + j] += dsn[j] * s->v[j];
            dp[L->o_c] += dsn[j] * dsg * s->sp[j];
            dsp[j] = dsn[j] * dsg * C;
            dpre2[j] = dsn[j] * p[L->o_z + j] * ((real)1 - s->v[j] * s->v[j]);
            dp[L->o_bh + j] += dpre2[j];
            for (int o = 0; o < APN_NODE; ++o) { dp[L->o_wh + j * APN_NODE + o] += dpre2[j] * s->v1[o]; dv1[o] += p[L->o_wh + j * APN_NODE + o] * dpre2[j]; }
        }
        for (int o = 0; o < APN_NODE; ++o) {
            dpre1[o] = dv1[o] * ((real)1 - s->v1[o] * s->v1[o]);
            dp[L->o_bu + o] += dpre1[o];
            for (int k = 0; k < 8; ++k) { dp[L->o_wu + o * (8 + n) + k] += dpre1[o] * s->feat[k]; dfeat[k] += p[L->o_wu + o * (8 + n) + k] * dpre1[o]; }
            for (int k = 0; k < n; ++k) { dp[L->o_wu + o * (8 + n) + 8 + k] += dpre1[o] * s->sp[k]; dsp[k] += p[L->o_wu + o * (8 + n) + 8 + k] * dpre1[o]; }
        }
        /* sp_I = h_I rr - h_Q ri,  sp_Q = h_I ri + h_Q rr (previous state into the normalised frame) */
        for (int j = 0; j < H; ++j) {
            dhI[j] = dsp[j] * s->rr + dsp[H + j] * s->ri; dhQ[j] = dsp[H + j] * s->rr - dsp[j] * s->ri;
            drr += dsp[j] * s->hI0[j] + dsp[H + j] * s->hQ0[j]; dri += dsp[H + j] * s->hI0[j] - dsp[j] * s->hQ0[j];
        }
        for (int j = 0; j < 3; ++j) dhA[j] = dsp[2 * H + j];
        /* feat_re = rr fi - ri fq,  feat_im = ri fi + rr fq */
        real dfi[4], dfq[4];
        for (int k = 0; k < 4; ++k) {
            dfi[k] = dfeat[2 * k] * s->rr + dfeat[2 * k + 1] * s->ri; dfq[k] = dfeat[2 * k + 1] * s->rr - dfeat[2 * k] * s->ri;
            drr += dfeat[2 * k] * s->fi[k] + dfeat[2 * k + 1] * s->fq[k]; dri += dfeat[2 * k + 1] * s->fi[k] - dfeat[2 * k] * s->fq[k];
        }
        for (int q = 0; q < APN_F; ++q)
            for (int m = 0; m < APN_M; ++m) {
                const int tt = t - (APN_M - 1) + m;
                if (tt < 0) continue;
                const real xi = x[2 * tt], xq = x[2 * tt + 1], bi = p[L->o_bi + q * APN_M + m], bq = p[L->o_bq + q * APN_M + m];
                dp[L->o_bi + q * APN_M + m] += dfi[q] * xi + dfq[q] * xq; dp[L->o_bq + q * APN_M + m] += dfq[q] * xi - dfi[q] * xq;
                if (dx) { dx[2 * tt] += dfi[q] * bi + dfq[q] * bq; dx[2 * tt + 1] += dfq[q] * bi - dfi[q] * bq; }
            }
        if (dx) {   /* the raw sample as the fourth value, and r = (I, -Q) / |x| */
            const real I = x[2 * t], Q = x[2 * t + 1], m3 = s->mag * s->mag * s->mag, w = (drr * Q + dri * I) / m3;
            dx[2 * t] += dfi[3] + Q * w; dx[2 * t + 1] += dfq[3] - I * w;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* MCLDNN: mcldnn.py:9-134.  Per step a 5 x 5 patch P[f][m]: features f = (I, Q, a, a^2, a^3) (:104-111) at the samples t-4+m, the
 * frame's own last four samples standing in front of it (circular window, :115-118).  conv2d_1 = Conv2d(1 -> C, 3x3, pad 1) on P
 * (:121);  conv1d = Conv1d(5 -> 5C, k3, pad 1, groups 5) along m, its (5C, 5) result RE-READ as (C, 5, 5): entry [c][f][m] is output
 * channel 5c + f, which belongs to input feature (5c + f) / C (:122-123);  both stacked to 10 "channels" x (C x 5) (:124-126) and merged
 * by conv2d_2 = Conv2d(10 -> 1, 3x3, pad 1) over (C, 5) -> z (5C, index 5c + m) (:126-127);  nn.LSTM(5C -> 8) from the zero state (h_0 is
 * not passed, :128);  fc_out (8 -> 16), fc_out_2 (16 -> 2), no activation in between (:129-131).  hidden = C (models.py:131-132).
 * Parameter order (named_parameters): conv2d_1 (C,1,3,3)+b, conv1d (5C,1,3)+b, conv2d_2 (1,10,3,3)+b, lstm W_ih (32,5C), W_hh (32,8),
 * b_ih, b_hh, fc_out (16,8)+b, fc_out_2 (2,16)+b. */
#define MCL_H 8
#define MCL_MAXC MAXH
typedef struct { int C; int64_t o_w1, o_b1, o_w1d, o_b1d, o_w2, o_b2, o_wih, o_whh, o_bih, o_bhh, o_wf1, o_bf1, o_wf2, o_bf2, P; } mcl_layout_t;
static void mcl_layout(const odpd_model_t* m, mcl_layout_t* g) {
    int64_t C = m->hidden, o = 0;
    g->C = (int)C;
    g->o_w1 = o; o += 9 * C; g->o_b1 = o; o += C;
    g->o_w1d = o; o += 15 * C; g->o_b1d = o; o += 5 * C;
    g->o_w2 = o; o += 90; g->o_b2 = o; o += 1;
    g->o_wih = o; o += 4 * MCL_H * 5 * C; g->o_whh = o; o += 4 * MCL_H * MCL_H; g->o_bih = o; o += 4 * MCL_H; g->o_bhh = o; o += 4 * MCL_H;
    g->o_wf1 = o; o += 16 * MCL_H; g->o_bf1 = o; o += 16; g->o_wf2 = o; o += 32; g->o_bf2 = o; o += 2;
    g->P = o;
}
typedef struct { real gi[MCL_H], gf[MCL_H], gg[MCL_H], go[MCL_H], cp[MCL_H], c[MCL_H], hp[MCL_H], h[MCL_H]; } mcl_step_t;
static void mcl_feat(const real* x, int s, real* f) {
    const real I = x[2 * s], Q = x[2 * s + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
    f[0] = I; f[1] = Q; f[2] = a; f[3] = a2; f[4] = a * a * a;
}
/* patch of step t and the front end up to z; cat[ch][c][m] (ch < 5: conv2d_1 feature row ch, ch >= 5: the re-read conv1d) */
static void mcl_front(const mcl_layout_t* L, const real* p, int T, const real* x, int t, real P[5][5], real* cat, real* z) {
    const int C = L->C;
    for (int m = 0; m < 5; ++m) {
        real f[5];
        mcl_feat(x, ((t - 4 + m) % T + T) % T, f);
        for (int k = 0; k < 5; ++k) P[k][m] = f[k];
    }
    for (int c = 0; c < C; ++c)
        for (int f = 0; f < 5; ++f)
            for (int m = 0; m < 5; ++m) {
                real a = p[L->o_b1 + c];
                for (int df = 0; df < 3; ++df)
                    for (int dm = 0; dm < 3; ++dm) {
                        const int ff = f + df - 1, mm = m + dm - 1;
                        if (ff >= 0 && ff < 5 && mm >= 0 && mm < 5) a += p[L->o_w1 + c * 9 + df * 3 + dm] * P[ff][mm];
                    }
                cat[(f * C + c) * 5 + m] = a;
            }
    for (int oc = 0; oc < 5 * C; ++oc)
        for (int m = 0; m < 5; ++m) {
            real a = p[L->o_b1d + oc];
            for (int dm = 0; dm < 3; ++dm) {
                const int mm = m + dm - 1;
                if (mm >= 0 && mm < 5) a += p[L->o_w1d + oc * 3 + dm] * P[oc / C][mm];
            }
            cat[((5 + oc % 5) * C + oc / 5) * 5 + m] = a;
        }
    for (int c = 0; c < C; ++c)
        for (int m = 0; m < 5; ++m) {
            real a = p[L->o_b2];
            for (int ch = 0; ch < 10; ++ch)
                for (int dc = 0; dc < 3; ++dc)
                    for (int dm = 0; dm < 3; ++dm) {
                        const int cc = c + dc - 1, mm = m + dm - 1;
                        if (cc >= 0 && cc < C && mm >= 0 && mm < 5) a += p[L->o_w2 + ch * 9 + dc * 3 + dm] * cat[(ch * C + cc) * 5 + mm];
                    }
            z[c * 5 + m] = a;
        }
}
static void mcl_seq_fwd(const mcl_layout_t* L, const real* p, int T, const real* x, real* y, mcl_step_t* S, real* scratch) {
    const int C = L->C, Z = 5 * C;
    real h[MCL_H] = {0}, c[MCL_H] = {0};
    real* cat = scratch; real* z = cat + 10 * C * 5;
    mcl_step_t tmp;
    for (int t = 0; t < T; ++t) {
        mcl_step_t* s = S ? &S[t] : &tmp;
        real P[5][5];
        mcl_front(L, p, T, x, t, P, cat, z);
        for (int j = 0; j < MCL_H; ++j) {
            real pre[4];
            for (int g = 0; g < 4; ++g) {
                real a = p[L->o_bih + g * MCL_H + j] + p[L->o_bhh + g * MCL_H + j];
                for (int k = 0; k < Z; ++k) a += p[L->o_wih + (g * MCL_H + j) * Z + k] * z[k];
                for (int k = 0; k < MCL_H; ++k) a += p[L->o_whh + (g * MCL_H + j) * MCL_H + k] * h[k];
                pre[g] = a;
            }
            s->gi[j] = sigm(pre[0]); s->gf[j] = sigm(pre[1]); s->gg[j] = tanhr(pre[2]); s->go[j] = sigm(pre[3]);
        }
        for (int j = 0; j < MCL_H; ++j) {
            s->cp[j] = c[j]; s->hp[j] = h[j];
            c[j] = s->gf[j] * c[j] + s->gi[j] * s->gg[j];
            s->c[j] = c[j];
        }
        for (int j = 0; j < MCL_H; ++j) { h[j] = s->go[j] * tanhr(c[j]); s->h[j] = h[j]; }
        real f1[16];
        for (int o = 0; o < 16; ++o) {
            real a = p[L->o_bf1 + o];
            for (int k = 0; k < MCL_H; ++k) a += p[L->o_wf1 + o * MCL_H + k] * h[k];
            f1[o] = a;
        }
        for (int o = 0; o < 2; ++o) {
            real a = p[L->o_bf2 + o];
            for (int k = 0; k < 16; ++k) a += p[L->o_wf2 + o * 16 + k] * f1[k];
            y[2 * t + o] = a;
        }
    }
}
static void mcl_seq_bwd(const mcl_layout_t* L, const real* p, int T, const real* x, const real* dy, const mcl_step_t* S, real* dp, real* dx,
                        real* scratch) {
    const int C = L->C, Z = 5 * C;
    real* cat = scratch; real* z = cat + 10 * C * 5; real* dcat = z + Z; real* dz = dcat + 10 * C * 5; real* dfeat = dz + Z;   /* dfeat: T x 5 */
    real dh[MCL_H] = {0}, dc[MCL_H] = {0};
    for (int i = 0; i < 5 * T; ++i) dfeat[i] = 0;
    for (int t = T - 1; t >= 0; --t) {
        const mcl_step_t* s = &S[t];
        real P[5][5], f1[16], df1[16] = {0}, dpre[4][MCL_H], nh[MCL_H] = {0};
        mcl_front(L, p, T, x, t, P, cat, z);
        for (int o = 0; o < 16; ++o) {
            real a = p[L->o_bf1 + o];
            for (int k = 0; k < MCL_H; ++k) a += p[L->o_wf1 + o * MCL_H + k] * s->h[k];
            f1[o] = a;
        }
        for (int o = 0; o < 2; ++o) {
            dp[L->o_bf2 + o] += dy[2 * t + o];
            for (int k = 0; k < 16; ++k) { dp[L->o_wf2 + o * 16 + k] += dy[2 * t + o] * f1[k]; df1[k] += p[L->o_wf2 + o * 16 + k] * dy[2 * t + o]; }
        }
        for (int o = 0; o < 16; ++o) {
            dp[L->o_bf1 + o] += df1[o];
            for (int k = 0; k < MCL_H; ++k) { dp[L->o_wf1 + o * MCL_H + k] += df1[o] * s->h[k]; dh[k] += p[L->o_wf1 + o * MCL_H + k] * df1[o]; }
        }
        for (int j = 0; j < MCL_H; ++j) {
            const real tc = tanhr(s->c[j]);
            const real dct = dc[j] + dh[j] * s->go[j] * ((real)1 - tc * tc);
            dpre[3][j] = dh[j] * tc * s->go[j] * ((real)1 - s->go[j]);
            dpre[0][j] = dct * s->gg[j] * s->gi[j] * ((real)1 - s->gi[j]);
            dpre[1][j] = dct * s->cp[j] * s->gf[j] * ((real)1 - s->gf[j]);
            dpre[2][j] = dct * s->gi[j] * ((real)1 - s->gg[j] * s->gg[j]);
            dc[j] = dct * s->gf[j];
        }
        for (int k = 0; k < Z; ++k) dz[k] = 0;
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < MCL_H; ++j) {
                const real d = dpre[g][j];
                dp[L->o_bih + g * MCL_H + j] += d; dp[L->o_bhh + g * MCL_H + j] += d;
                for (int k = 0; k < Z; ++k) { dp[L->o_wih + (g * MCL_H + j) * Z + k] += d * z[k]; dz[k] += p[L->o_wih + (g * MCL_H + j) * Z + k] * d; }
                for (int k = 0; k < MCL_H; ++k) { dp[L->o_whh + (g * MCL_H + j) * MCL_H + k] += d * s->hp[k]; nh[k] += p[L->o_whh + (g * MCL_H + j) * MCL_H + k] * d; }
            }
        for (int j = 0; j < MCL_H; ++j) dh[j] = nh[j];
        /* conv2d_2 */
        for (int i = 0; i < 10 * C * 5; ++i) dcat[i] = 0;
        for (int c = 0; c < C; ++c)
            for (int m = 0; m < 5; ++m) {
                const real d = dz[c * 5 + m];
                dp[L->o_b2] += d;
                for (int ch = 0; ch < 10; ++ch)
                    for (int dcc = 0; dcc < 3; ++dcc)
                        for (int dm = 0; dm < 3; ++dm) {
                            const int cc = c + dcc - 1, mm = m + dm - 1;
                            if (cc >= 0 && cc < C && mm >= 0 && mm < 5) {
                                dp[L->o_w2 + ch * 9 + dcc * 3 + dm] += d * cat[(ch * C + cc) * 5 + mm];
                                dcat[(ch * C + cc) * 5 + mm] += p[L->o_w2 + ch * 9 + dcc * 3 + dm] * d;
                            }
                        }
            }
        real dP[5][5] = {{0}};
        for (int c = 0; c < C; ++c)
            for (int f = 0; f < 5; ++f)
                for (int m = 0; m < 5; ++m) {
                    const real d = dcat[(f * C + c) * 5 + m];
                    dp[L->o_b1 + c] += d;
                    for (int df = 0; df < 3; ++df)
                        for (int dm = 0; dm < 3; ++dm) {
                            const int ff = f + df - 1, mm = m + dm - 1;
                            if (ff >= 0 && ff < 5 && mm >= 0 && mm < 5) { dp[L->o_w1 + c * 9 + df * 3 + dm] += d * P[ff][mm]; dP[ff][mm] += p[L->o_w1 + c * 9 + df * 3 + dm] * d; }
                        }
                }
        for (int oc = 0; oc < 5 * C; ++oc)
            for (int m = 0; m < 5; ++m) {
                const real d = dcat[((5 + oc % 5) * C + oc / 5) * 5 + m];
                dp[L->o_b1d + oc] += d;
                for (int dm = 0; dm < 3; ++dm) {
                    const int mm = m + dm - 1;
                    if (mm >= 0 && mm < 5) { dp[L->o_w1d + oc * 3 + dm] += d * P[oc / C][mm]; dP[oc / C][mm] += p[L->o_w1d + oc * 3 + dm] * d; }
                }
            }
        for (int m = 0; m < 5; ++m) {
            const int sidx = ((t - 4 + m) % T + T) % T;
            for (int k = 0; k < 5; ++k) dfeat[sidx * 5 + k] += dP[k][m];
        }
    }
    if (dx)
        for (int t = 0; t < T; ++t) {   /* features (I, Q, a, a^2, a^3) */
            const real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
            const real* d = dfeat + 5 * t;
            const real da = d[2] + (real)2 * a * d[3] + (real)3 * a2 * d[4];
            dx[2 * t] = d[0] + da * I / a; dx[2 * t + 1] = d[1] + da * Q / a;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Quantisation-aware models: quant/__init__.py:20-37 -> quant_envs.py:138-306, the GENERIC surgery */
/*   (nn.GRU -> Python GRU of GRUCells :114-130; every Sigmoid / Tanh / Add / Mul module and every   */
/*   nn.Linear swapped :290-306) applied to gru.py, dgru.py, qgru.py, qgru_amp1.py (GRUCell,         */
/*   quant/modules/gru.py:43-59) and to deltagru_tcnskip.py (its DeltaGRULayer :156-162, 266-291).    */
/*   INT_Quantizer (quantizers.py:15-85): s = 2^round(log2|scale|); q(x) = round(clamp(x/s,Qn,Qp))*s */
/*   (clamp BEFORE round, round half to even), straight-through gradient inside the clamp range.   */
/*   INT_Linear (quant_layers.py:48-85): F.linear(q_a(x), q_w(W), b), bias not quantised; the       */
/*   16-bit out_quantizer applies to fc_out in eval mode only.                                      */
/* Parameter order = named_parameters() of the quantised model:                                    */
/*   GRUCell models: x2h.{weight,bias,wq.scale,aq.scale,oq.scale}, h2h.{same}, sigmoid.q, tanh.q,   */
/*     add.q, mul.q, fc_out.{weight,bias,wq,aq,oq} [, dgru: fc_hid.{weight,bias,wq,aq,oq}]           */
/*   deltagru_tcnskip: x2h.{weight,wq,aq,oq}, h2h.{same}, add.q, mul.q, sigmoid.q, tanh.q,           */
/*     fc_out.{weight,wq,aq,oq}, tcn.0.weight, tcn.2.weight                                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int H, F, OW, bits_w, bits_a, dgru, tres;
    int64_t o_wx, o_bx, o_sxw, o_sxa, o_sxo, o_wh, o_bh, o_shw, o_sha, o_sho, o_ssig, o_stanh, o_sadd, o_smul,
            o_wo, o_bo, o_sow, o_soa, o_soo, o_whid, o_bhid, o_shidw, o_shida, o_shido, o_tcn0, o_tcn2, P;
} qgru_layout_t;
static int is_qat(const odpd_model_t* m) {
    return m->bits_w > 0 && (m->backbone == ODPD_QGRU || m->backbone == ODPD_QGRU_AMP1 || m->backbone == ODPD_GRU ||
                             m->backbone == ODPD_DGRU || m->backbone == ODPD_TRES_DELTAGRU);
}
static void qgru_layout(const odpd_model_t* m, qgru_layout_t* g) {
    int64_t H = m->hidden, F = feat_dim(m->backbone), o = 0;
    memset(g, 0xff, sizeof(*g));
    g->H = (int)H; g->F = (int)F; g->bits_w = m->bits_w; g->bits_a = m->bits_a;
    g->dgru = m->backbone == ODPD_DGRU; g->tres = m->backbone == ODPD_TRES_DELTAGRU;
    g->OW = (int)(g->dgru ? H + 6 : H);
    if (g->tres) {
        g->o_wx = o; o += 3 * H * F; g->o_sxw = o++; g->o_sxa = o++; g->o_sxo = o++;
        g->o_wh = o; o += 3 * H * H; g->o_shw = o++; g->o_sha = o++; g->o_sho = o++;
        g->o_sadd = o++; g->o_smul = o++; g->o_ssig = o++; g->o_stanh = o++;
        g->o_wo = o; o += 2 * H; g->o_sow = o++; g->o_soa = o++; g->o_soo = o++;
        g->o_tcn0 = o; o += 18; g->o_tcn2 = o; o += 6;
    } else {
        g->o_wx = o; o += 3 * H * F; g->o_bx = o; o += 3 * H; g->o_sxw = o++; g->o_sxa = o++; g->o_sxo = o++;
        g->o_wh = o; o += 3 * H * H; g->o_bh = o; o += 3 * H; g->o_shw = o++; g->o_sha = o++; g->o_sho = o++;
        g->o_ssig = o++; g->o_stanh = o++; g->o_sadd = o++; g->o_smul = o++;
        g->o_wo = o; o += 2 * g->OW; g->o_bo = o; o += 2; g->o_sow = o++; g->o_soa = o++; g->o_soo = o++;
        if (g->dgru) { g->o_whid = o; o += H * H; g->o_bhid = o; o += H; g->o_shidw = o++; g->o_shida = o++; g->o_shido = o++; }
    }
    g->P = o;
}
static int64_t qat_param_count(const odpd_model_t* m) { qgru_layout_t L; qgru_layout(m, &L); return L.P; }
typedef struct {
    real f[MAXF], fq[MAXF], px[MAXF], hq[MAXH], ph[MAXH];  /* features, q_a(features), pass masks of act quantisers, q_a(h) */
    real hp[MAXH], xt[3 * MAXH], ht[3 * MAXH];
    real p_ar[MAXH], p_az[MAXH], p_an[MAXH], p_ah[MAXH];  /* pass masks of the four add-quantiser uses */
    real rf[MAXH], zf[MAXH], nf[MAXH];                    /* float sigmoid/tanh outputs (autograd saves these) */
    real p_r[MAXH], p_z[MAXH], p_n[MAXH];
    real r[MAXH], z[MAXH], n[MAXH];
    real p_m1[MAXH], p_m2[MAXH], p_m3[MAXH];
    real h[MAXH], ho[MAXH + 6], p_ho[MAXH + 6];           /* new state, q_a(fc_out input) and its mask */
    real h2[MAXH], p_h2[MAXH], hidpre[MAXH];              /* dgru: q_a(h') of fc_hid, its mask, fc_hid pre-activation */
    /* delta cell (deltagru_tcnskip) */
    real mx[MAXF], dxm[MAXF], mh[MAXH], dhm[MAXH], dmnh[MAXH], omz[MAXH], p_omz[MAXH], s1[3], s2[2];
} qgru_step_t;

static void qgru_seq_fwd(const odpd_model_t* m, const qgru_layout_t* L, const real* p, int T, const real* x, real* y,
                         qgru_step_t* S, int eval_mode, const real* qwx, const real* qwh, const real* qwo, const real* qwhid) {
    int H = L->H, F = L->F, ba = L->bits_a, OW = L->OW;
    real sxa = q_pow2(p[L->o_sxa]), sha = q_pow2(p[L->o_sha]), soa = q_pow2(p[L->o_soa]), soo = q_pow2(p[L->o_soo]);
    real ssig = q_pow2(p[L->o_ssig]), stanh = q_pow2(p[L->o_stanh]), sadd = q_pow2(p[L->o_sadd]), smul = q_pow2(p[L->o_smul]);
    real shida = L->dgru ? q_pow2(p[L->o_shida]) : (real)1;
    real h[MAXH] = {0};
    qgru_step_t tmp;
    for (int t = 0; t < T; ++t) {
        qgru_step_t* s = S ? &S[t] : &tmp;
        feat_fwd(m->backbone, x[2 * t], x[2 * t + 1], s->f);
        for (int i = 0; i < F; ++i) s->fq[i] = q_apply(s->f[i], sxa, ba, &s->px[i]);
        for (int j = 0; j < H; ++j) { s->hp[j] = h[j]; s->hq[j] = q_apply(h[j], sha, ba, &s->ph[j]); }
        for (int k = 0; k < 3 * H; ++k) {   /* exact grid sum, then the fp32 bias (matches F.linear bit for bit) */
            double a = 0, b = 0;
            for (int i = 0; i < F; ++i) a += (double)s->fq[i] * (double)qwx[k * F + i];
            for (int i = 0; i < H; ++i) b += (double)s->hq[i] * (double)qwh[k * H + i];
            s->xt[k] = (real)a + p[L->o_bx + k];
            s->ht[k] = (real)b + p[L->o_bh + k];
        }
        for (int j = 0; j < H; ++j) {
            real ar = q_apply(s->xt[j] + s->ht[j], sadd, ba, &s->p_ar[j]);
            real az = q_apply(s->xt[H + j] + s->ht[H + j], sadd, ba, &s->p_az[j]);
            s->rf[j] = sigm(ar); s->zf[j] = sigm(az);
            s->r[j] = q_apply(s->rf[j], ssig, ba, &s->p_r[j]);
            s->z[j] = q_apply(s->zf[j], ssig, ba, &s->p_z[j]);
            real m1 = q_apply(s->r[j] * s->ht[2 * H + j], smul, ba, &s->p_m1[j]);
            real an = q_apply(s->xt[2 * H + j] + m1, sadd, ba, &s->p_an[j]);
            s->nf[j] = tanhr(an);
            s->n[j] = q_apply(s->nf[j], stanh, ba, &s->p_n[j]);
            real m2 = q_apply(s->z[j] * h[j], smul, ba, &s->p_m2[j]);
            real m3 = q_apply(((real)1 - s->z[j]) * s->n[j], smul, ba, &s->p_m3[j]);
            s->h[j] = q_apply(m2 + m3, sadd, ba, &s->p_ah[j]);
        }
        for (int j = 0; j < H; ++j) h[j] = s->h[j];
        if (L->dgru) {   /* dgru.py:70-73 with fc_hid / fc_out as INT_Linear: relu(fc_hid(out)), cat with the float features */
            for (int j = 0; j < H; ++j) s->h2[j] = q_apply(h[j], shida, ba, &s->p_h2[j]);
            real cat[MAXH + 6];
            for (int j = 0; j < H; ++j) {
                double a = 0;
                for (int k = 0; k < H; ++k) a += (double)s->h2[k] * (double)qwhid[j * H + k];
                s->hidpre[j] = (real)a + p[L->o_bhid + j];
                cat[j] = s->hidpre[j] > 0 ? s->hidpre[j] : (real)0;
            }
            for (int i = 0; i < 6; ++i) cat[H + i] = s->f[i];
            for (int k = 0; k < OW; ++k) s->ho[k] = q_apply(cat[k], soa, ba, &s->p_ho[k]);
        } else {
            for (int j = 0; j < H; ++j) s->ho[j] = q_apply(h[j], soa, ba, &s->p_ho[j]);
        }
        for (int c = 0; c < 2; ++c) {
            double a = 0;
            for (int j = 0; j < OW; ++j) a += (double)s->ho[j] * (double)qwo[c * OW + j];
            real v = (real)a + p[L->o_bo + c];
            y[2 * t + c] = eval_mode ? q_apply(v, soo, 16, NULL) : v;
        }
    }
}
static void qgru_seq_bwd(const odpd_model_t* m, const qgru_layout_t* L, const real* p, int T, const real* x, const real* dy,
                         const qgru_step_t* S, real* dp, real* dx, const real* qwx, const real* qwh, const real* qwo,
                         const real* qwhid, real* dqwx, real* dqwh, real* dqwo, real* dqwhid) {
    int H = L->H, F = L->F, OW = L->OW;
    real dh[MAXH] = {0};
    for (int t = T - 1; t >= 0; --t) {
        const qgru_step_t* s = &S[t];
        real dhn[MAXH];   /* gradient w.r.t. the new state h' */
        real df[MAXF] = {0};
        for (int j = 0; j < H; ++j) dhn[j] = dh[j];
        real dcat[MAXH + 6] = {0};
        for (int c = 0; c < 2; ++c) {
            real d = dy[2 * t + c];
            dp[L->o_bo + c] += d;
            for (int j = 0; j < OW; ++j) { dqwo[c * OW + j] += d * s->ho[j]; dcat[j] += d * qwo[c * OW + j] * s->p_ho[j]; }
        }
        if (L->dgru) {
            for (int i = 0; i < 6; ++i) df[i] += dcat[H + i];
            for (int j = 0; j < H; ++j) {
                real dpre = s->hidpre[j] > 0 ? dcat[j] : (real)0;
                dp[L->o_bhid + j] += dpre;
                for (int k = 0; k < H; ++k) { dqwhid[j * H + k] += dpre * s->h2[k]; dhn[k] += dpre * qwhid[j * H + k] * s->p_h2[k]; }
            }
        } else {
            for (int j = 0; j < H; ++j) dhn[j] += dcat[j];
        }
        real dxt[3 * MAXH], dht[3 * MAXH], dhp[MAXH];
        for (int j = 0; j < H; ++j) {
            real g = dhn[j] * s->p_ah[j];              /* through q_add(m2 + m3) */
            real dm2 = g * s->p_m2[j], dm3 = g * s->p_m3[j];
            real dz = dm2 * s->hp[j] - dm3 * s->n[j];
            dhp[j] = dm2 * s->z[j];
            real dn = dm3 * ((real)1 - s->z[j]);
            real dan = dn * s->p_n[j] * ((real)1 - s->nf[j] * s->nf[j]) * s->p_an[j];
            dxt[2 * H + j] = dan;
            real dm1 = dan * s->p_m1[j];
            real dr = dm1 * s->ht[2 * H + j];
            dht[2 * H + j] = dm1 * s->r[j];
            real dar = dr * s->p_r[j] * s->rf[j] * ((real)1 - s->rf[j]) * s->p_ar[j];
            real daz = dz * s->p_z[j] * s->zf[j] * ((real)1 - s->zf[j]) * s->p_az[j];
            dxt[j] = dar; dht[j] = dar; dxt[H + j] = daz; dht[H + j] = daz;
        }
        for (int k = 0; k < 3 * H; ++k) {
            dp[L->o_bx + k] += dxt[k]; dp[L->o_bh + k] += dht[k];
            for (int i = 0; i < F; ++i) {
                dqwx[k * F + i] += dxt[k] * s->fq[i];
                df[i] += dxt[k] * qwx[k * F + i] * s->px[i];
            }
            for (int i = 0; i < H; ++i) { dqwh[k * H + i] += dht[k] * s->hq[i]; dhp[i] += dht[k] * qwh[k * H + i] * s->ph[i]; }
        }
        for (int j = 0; j < H; ++j) dh[j] = dhp[j];
        if (dx) feat_bwd(m->backbone, x[2 * t], x[2 * t + 1], df, &dx[2 * t], &dx[2 * t + 1]);
    }
}

/* deltagru_tcnskip under the surgery: x2h / h2h bias-free INT_Linear on the thresholded deltas, accumulators in fp32,
 * gate_r = Qsig(dm_r), gate_z = Qsig(dm_z), gate_n = Qtanh(Qadd(dm_n, Qmul(gate_r, dm_nh))),
 * h = Qadd(Qmul(Qadd(1, -gate_z), gate_n), Qmul(gate_z, h))  (deltagru_tcnskip.py:266-291), y = fc_out(h) [eval: 16-bit] + skip */
static void qtres_seq_fwd(const odpd_model_t* m, const qgru_layout_t* L, const real* p, int T, const real* x, real* y,
                          qgru_step_t* S, int eval_mode, const real* qwx, const real* qwh, const real* qwo, double* stats) {
    int H = L->H, ba = L->bits_a;
    const real thx = (real)(float)m->thx, thh = (real)(float)m->thh;
    real sxa = q_pow2(p[L->o_sxa]), sha = q_pow2(p[L->o_sha]), soa = q_pow2(p[L->o_soa]), soo = q_pow2(p[L->o_soo]);
    real ssig = q_pow2(p[L->o_ssig]), stanh = q_pow2(p[L->o_stanh]), sadd = q_pow2(p[L->o_sadd]), smul = q_pow2(p[L->o_smul]);
    real xp[6] = {0}, h[MAXH] = {0}, hp[MAXH] = {0}, dm[3 * MAXH] = {0}, dmnh[MAXH] = {0};
    delta_layout_t DL; DL.tres = 1; DL.H = H;
    qgru_step_t tmp;
    double zx = 0, zh = 0;
    for (int t = 0; t < T; ++t) {
        qgru_step_t* s = S ? &S[t] : &tmp;
        delta_feat(&DL, x, t, T, s->f);
        for (int i = 0; i < 6; ++i) {
            real d = s->f[i] - xp[i], ad = (real)fabs((double)d);
            s->mx[i] = (ad < thx) ? (real)0 : (real)1;
            s->dxm[i] = s->mx[i] != 0 ? d : (real)0;
            if (s->dxm[i] == 0) zx += 1;
            if (ad >= thx) xp[i] = s->f[i];
            s->fq[i] = q_apply(s->dxm[i], sxa, ba, &s->px[i]);
        }
        for (int j = 0; j < H; ++j) {
            real d = h[j] - hp[j], ad = (real)fabs((double)d);
            s->hp[j] = h[j];
            s->mh[j] = (ad < thh) ? (real)0 : (real)1;
            s->dhm[j] = s->mh[j] != 0 ? d : (real)0;
            if (s->dhm[j] == 0) zh += 1;
            if (ad >= thh) hp[j] = h[j];
            s->hq[j] = q_apply(s->dhm[j], sha, ba, &s->ph[j]);
        }
        for (int j = 0; j < H; ++j) {
            real mx[3], mh[3];
            for (int k = 0; k < 3; ++k) {
                double a = 0, b = 0;
                for (int i = 0; i < 6; ++i) a += (double)s->fq[i] * (double)qwx[(k * H + j) * 6 + i];
                for (int i = 0; i < H; ++i) b += (double)s->hq[i] * (double)qwh[(k * H + j) * H + i];
                mx[k] = (real)a + dm[k * H + j]; mh[k] = (real)b;
            }
            dm[j] = mx[0] + mh[0]; dm[H + j] = mx[1] + mh[1]; dm[2 * H + j] = mx[2];
            dmnh[j] = mh[2] + dmnh[j];
            s->dmnh[j] = dmnh[j];
            s->rf[j] = sigm(dm[j]); s->zf[j] = sigm(dm[H + j]);
            s->r[j] = q_apply(s->rf[j], ssig, ba, &s->p_r[j]);
            s->z[j] = q_apply(s->zf[j], ssig, ba, &s->p_z[j]);
            real m1 = q_apply(s->r[j] * dmnh[j], smul, ba, &s->p_m1[j]);
            real an = q_apply(dm[2 * H + j] + m1, sadd, ba, &s->p_an[j]);
            s->nf[j] = tanhr(an);
            s->n[j] = q_apply(s->nf[j], stanh, ba, &s->p_n[j]);
            s->omz[j] = q_apply((real)1 + (-s->z[j]), sadd, ba, &s->p_omz[j]);
            real m3 = q_apply(s->omz[j] * s->n[j], smul, ba, &s->p_m3[j]);
            real m2 = q_apply(s->z[j] * h[j], smul, ba, &s->p_m2[j]);
            s->h[j] = q_apply(m3 + m2, sadd, ba, &s->p_ah[j]);
        }
        for (int j = 0; j < H; ++j) { h[j] = s->h[j]; s->ho[j] = q_apply(h[j], soa, ba, &s->p_ho[j]); }
        for (int c = 0; c < 2; ++c) {
            double a = 0;
            for (int j = 0; j < H; ++j) a += (double)s->ho[j] * (double)qwo[c * H + j];
            y[2 * t + c] = eval_mode ? q_apply((real)a, soo, 16, NULL) : (real)a;
        }
        for (int c = 0; c < 3; ++c) {   /* the TCN skip stays float (Conv1d / Hardswish are not swapped) */
            real a = 0;
            for (int i = 0; i < 2; ++i)
                for (int k = 0; k < 3; ++k) {
                    int tt = t + 16 * (k - 1);
                    if (tt >= 0 && tt < T) a += p[L->o_tcn0 + (c * 2 + i) * 3 + k] * x[2 * tt + i];
                }
            s->s1[c] = a;
        }
        for (int o = 0; o < 2; ++o) {
            real a = 0;
            for (int c = 0; c < 3; ++c) a += p[L->o_tcn2 + o * 3 + c] * hswish(s->s1[c]);
            s->s2[o] = a;
            y[2 * t + o] += hswish(a);
        }
    }
    if (stats) { stats[0] += zx; stats[1] += 6.0 * T; stats[2] += zh; stats[3] += (double)H * T; }
}
static void qtres_seq_bwd(const odpd_model_t* m, const qgru_layout_t* L, const real* p, int T, const real* x, const real* dy,
                          const qgru_step_t* S, real* dp, real* dx, const real* qwx, const real* qwh, const real* qwo,
                          real* dqwx, real* dqwh, real* dqwo) {
    int H = L->H;
    (void)m;
    real Gh[MAXH] = {0}, Ghp[MAXH] = {0}, Gxp[6] = {0}, Gdm[3 * MAXH] = {0}, Gnh[MAXH] = {0};
    real* dfeat = (real*)calloc((size_t)T * 6, sizeof(real));
    if (dx) memset(dx, 0, sizeof(real) * 2 * T);
    for (int t = T - 1; t >= 0; --t) {
        const qgru_step_t* s = &S[t];
        for (int c = 0; c < 2; ++c) {
            real d = dy[2 * t + c];
            for (int j = 0; j < H; ++j) { dqwo[c * H + j] += d * s->ho[j]; Gh[j] += d * qwo[c * H + j] * s->p_ho[j]; }
        }
        {
            real dh1[3] = {0};
            for (int o = 0; o < 2; ++o) {
                real d2 = dy[2 * t + o] * hswish_grad(s->s2[o]);
                for (int c = 0; c < 3; ++c) { dp[L->o_tcn2 + o * 3 + c] += d2 * hswish(s->s1[c]); dh1[c] += d2 * p[L->o_tcn2 + o * 3 + c]; }
            }
            for (int c = 0; c < 3; ++c) {
                real d1 = dh1[c] * hswish_grad(s->s1[c]);
                for (int i = 0; i < 2; ++i)
                    for (int k = 0; k < 3; ++k) {
                        int tt = t + 16 * (k - 1);
                        if (tt >= 0 && tt < T) {
                            dp[L->o_tcn0 + (c * 2 + i) * 3 + k] += d1 * x[2 * tt + i];
                            if (dx) dx[2 * tt + i] += d1 * p[L->o_tcn0 + (c * 2 + i) * 3 + k];
                        }
                    }
            }
        }
        real Ghprev[MAXH];
        for (int j = 0; j < H; ++j) {
            real g = Gh[j] * s->p_ah[j];
            real dm3 = g * s->p_m3[j], dm2 = g * s->p_m2[j];
            real domz = dm3 * s->n[j], dn = dm3 * s->omz[j];
            real dz = dm2 * s->hp[j] - domz * s->p_omz[j];
            Ghprev[j] = dm2 * s->z[j];
            real dan = dn * s->p_n[j] * ((real)1 - s->nf[j] * s->nf[j]) * s->p_an[j];
            Gdm[2 * H + j] += dan;
            real dm1 = dan * s->p_m1[j];
            real dr = dm1 * s->dmnh[j];
            Gnh[j] += dm1 * s->r[j];
            Gdm[j] += dr * s->p_r[j] * s->rf[j] * ((real)1 - s->rf[j]);
            Gdm[H + j] += dz * s->p_z[j] * s->zf[j] * ((real)1 - s->zf[j]);
        }
        real ddx[6] = {0}, ddh[MAXH] = {0};
        for (int j = 0; j < H; ++j)
            for (int k = 0; k < 3; ++k) {
                real gx = Gdm[k * H + j], gh = (k < 2) ? Gdm[k * H + j] : Gnh[j];
                for (int i = 0; i < 6; ++i) { dqwx[(k * H + j) * 6 + i] += gx * s->fq[i]; ddx[i] += gx * qwx[(k * H + j) * 6 + i]; }
                for (int i = 0; i < H; ++i) { dqwh[(k * H + j) * H + i] += gh * s->hq[i]; ddh[i] += gh * qwh[(k * H + j) * H + i]; }
            }
        for (int i = 0; i < 6; ++i) {
            real mk = s->mx[i], g = ddx[i] * s->px[i];
            dfeat[t * 6 + i] += mk * g + mk * Gxp[i];
            Gxp[i] = ((real)1 - mk) * Gxp[i] - mk * g;
        }
        for (int j = 0; j < H; ++j) {
            real mk = s->mh[j], g = ddh[j] * s->ph[j];
            Ghprev[j] += mk * g + mk * Ghp[j];
            Ghp[j] = ((real)1 - mk) * Ghp[j] - mk * g;
            Gh[j] = Ghprev[j];
        }
    }
    if (dx)
        for (int t = 0; t < T; ++t) {
            const real* df = dfeat + t * 6;
            real I = x[2 * t], Q = x[2 * t + 1], a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
            real da = df[2] + (real)3 * a * a * df[3];
            dx[2 * t] += df[0] + da * I / a; dx[2 * t + 1] += df[1] + da * Q / a;
            int tn = (t + 1) % T;
            dx[2 * tn] += df[4]; dx[2 * tn + 1] += df[5];
        }
    free(dfeat);
}
/* quantised weights and their pass masks (STE to the float master weights); order: x2h, h2h, fc_out, fc_hid */
static int qat_nq(const qgru_layout_t* L, int* nx, int* nh, int* no, int* nhid) {
    *nx = 3 * L->H * L->F; *nh = 3 * L->H * L->H; *no = 2 * L->OW; *nhid = L->dgru ? L->H * L->H : 0;
    return *nx + *nh + *no + *nhid;
}
static void qgru_quant_weights(const qgru_layout_t* L, const real* p, real* qw, real* mk) {
    int nx, nh, no, nhid; qat_nq(L, &nx, &nh, &no, &nhid);
    real sx = q_pow2(p[L->o_sxw]), sh = q_pow2(p[L->o_shw]), so = q_pow2(p[L->o_sow]);
    for (int i = 0; i < nx; ++i) qw[i] = q_apply(p[L->o_wx + i], sx, L->bits_w, mk ? &mk[i] : NULL);
    for (int i = 0; i < nh; ++i) qw[nx + i] = q_apply(p[L->o_wh + i], sh, L->bits_w, mk ? &mk[nx + i] : NULL);
    for (int i = 0; i < no; ++i) qw[nx + nh + i] = q_apply(p[L->o_wo + i], so, L->bits_w, mk ? &mk[nx + nh + i] : NULL);
    if (nhid) {
        real shd = q_pow2(p[L->o_shidw]);
        for (int i = 0; i < nhid; ++i) qw[nx + nh + no + i] = q_apply(p[L->o_whid + i], shd, L->bits_w, mk ? &mk[nx + nh + no + i] : NULL);
    }
}

/* forward of the quantised model; eval_mode != 0 applies fc_out's 16-bit output quantiser (quant_layers.py:77-80);
 * stats (nullable, 4 doubles) accumulates the delta cell's sparsity counters */
int oracle_qat_fwd(const odpd_model_t* m, int B, int T, const real* params, const real* x, real* y, int eval_mode, double* stats) {
    if (!m || !is_qat(m) || !params || !x || !y || B <= 0 || T <= 0 || m->hidden > MAXH) return ODPD_EINVAL;
    qgru_layout_t L; qgru_layout(m, &L);
    int nx, nh, no, nhid, nq = qat_nq(&L, &nx, &nh, &no, &nhid);
    real* qw = (real*)malloc(sizeof(real) * nq);
    real *qwx = qw, *qwh = qw + nx, *qwo = qwh + nh, *qwhid = qwo + no;
    qgru_quant_weights(&L, params, qw, NULL);
    double st[4] = {0, 0, 0, 0};
#pragma omp parallel
    {
        double sl[4] = {0, 0, 0, 0};
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            if (L.tres) qtres_seq_fwd(m, &L, params, T, x + (int64_t)b * T * 2, y + (int64_t)b * T * 2, NULL, eval_mode, qwx, qwh, qwo, sl);
            else qgru_seq_fwd(m, &L, params, T, x + (int64_t)b * T * 2, y + (int64_t)b * T * 2, NULL, eval_mode, qwx, qwh, qwo, qwhid);
        }
#pragma omp critical
        for (int i = 0; i < 4; ++i) st[i] += sl[i];
    }
    if (stats) for (int i = 0; i < 4; ++i) stats[i] += st[i];
    free(qw);
    return 0;
}
/* train-mode backward; dparams (P) overwritten: scale entries get exactly 0 (round() kills their gradient) */
int oracle_qat_bwd(const odpd_model_t* m, int B, int T, const real* params, const real* x, const real* dy, real* dparams, real* dx) {
    if (!m || !is_qat(m) || !params || !x || !dy || !dparams || B <= 0 || T <= 0 || m->hidden > MAXH) return ODPD_EINVAL;
    qgru_layout_t L; qgru_layout(m, &L);
    int nx, nh, no, nhid, nq = qat_nq(&L, &nx, &nh, &no, &nhid);
    int64_t P = L.P;
    memset(dparams, 0, sizeof(real) * P);
    real* qw = (real*)malloc(sizeof(real) * 2 * nq);
    real *qwx = qw, *qwh = qw + nx, *qwo = qwh + nh, *qwhid = qwo + no, *mk = qw + nq;
    qgru_quant_weights(&L, params, qw, mk);
    real* dq = (real*)calloc(nq, sizeof(real));
#pragma omp parallel
    {
        real* dp = (real*)calloc(P, sizeof(real));
        real* dql = (real*)calloc(nq, sizeof(real));
        qgru_step_t* S = (qgru_step_t*)malloc(sizeof(qgru_step_t) * T);
        real* ytmp = (real*)malloc(sizeof(real) * 2 * T);
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            const real* xb = x + (int64_t)b * T * 2;
            real* dxb = dx ? dx + (int64_t)b * T * 2 : NULL;
            if (L.tres) {
                qtres_seq_fwd(m, &L, params, T, xb, ytmp, S, 0, qwx, qwh, qwo, NULL);
                qtres_seq_bwd(m, &L, params, T, xb, dy + (int64_t)b * T * 2, S, dp, dxb, qwx, qwh, qwo, dql, dql + nx, dql + nx + nh);
            } else {
                qgru_seq_fwd(m, &L, params, T, xb, ytmp, S, 0, qwx, qwh, qwo, qwhid);
                qgru_seq_bwd(m, &L, params, T, xb, dy + (int64_t)b * T * 2, S, dp, dxb, qwx, qwh, qwo, qwhid,
                             dql, dql + nx, dql + nx + nh, dql + nx + nh + no);
            }
        }
#pragma omp critical
        {
            for (int64_t i = 0; i < P; ++i) dparams[i] += dp[i];
            for (int i = 0; i < nq; ++i) dq[i] += dql[i];
        }
        free(dp); free(dql); free(S); free(ytmp);
    }
    for (int i = 0; i < nx; ++i) dparams[L.o_wx + i] = dq[i] * mk[i];
    for (int i = 0; i < nh; ++i) dparams[L.o_wh + i] = dq[nx + i] * mk[nx + i];
    for (int i = 0; i < no; ++i) dparams[L.o_wo + i] = dq[nx + nh + i] * mk[nx + nh + i];
    for (int i = 0; i < nhid; ++i) dparams[L.o_whid + i] = dq[nx + nh + no + i] * mk[nx + nh + no + i];
    free(qw); free(dq);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* GMP: gmp.py:18-50.  u = x delayed by M-1 (zero history, :26-27); amp = |u| delayed by M-1 once more (:33);
 * per sample t the basis is [u[t+m]] (M terms) followed by u[t+m] * amp[t+i+m]^d for d = 1..D-1, i, m = 0..M-1
 * (:41-44, flattened d-major, then i, then m); y[t] = sum basis * Weight (real weights on complex terms, :46-48).  */
/* ------------------------------------------------------------------------------------------ */
static inline real gmp_pow(real a, int d) { real a2 = a * a; return d == 1 ? a : d == 2 ? a2 : d == 3 ? a2 * a : a2 * a2; }
static void gmp_windows(int M, int T, const real* x, real* u, real* amp) {
    for (int k = 0; k < T + M - 1; ++k) {
        int s = k - (M - 1);
        u[2 * k] = s >= 0 ? x[2 * s] : (real)0;
        u[2 * k + 1] = s >= 0 ? x[2 * s + 1] : (real)0;
    }
    for (int k = 0; k < T + 2 * M - 2; ++k) {
        int s = k - (M - 1);
        amp[k] = s >= 0 ? (real)sqrt((double)u[2 * s] * u[2 * s] + (double)u[2 * s + 1] * u[2 * s + 1]) : (real)0;
    }
}
static void gmp_seq_fwd(int M, const real* w, int T, const real* x, real* y, real* u, real* amp) {
    gmp_windows(M, T, x, u, amp);
    for (int t = 0; t < T; ++t) {
        real yr = 0, yi = 0;
        for (int m = 0; m < M; ++m) {
            real e = w[m];
            for (int d = 1; d < GMP_DEGREE; ++d)
                for (int i = 0; i < M; ++i) e += w[M + ((d - 1) * M + i) * M + m] * gmp_pow(amp[t + i + m], d);
            yr += e * u[2 * (t + m)];
            yi += e * u[2 * (t + m) + 1];
        }
        y[2 * t] = yr; y[2 * t + 1] = yi;
    }
}
/* du / damp are scratch of T+M-1 complex / T+2M-2 real entries */
static void gmp_seq_bwd(int M, const real* w, int T, const real* x, const real* dy, real* dp, real* dx, real* u, real* amp,
                        real* du, real* damp) {
    gmp_windows(M, T, x, u, amp);
    memset(du, 0, sizeof(real) * 2 * (T + M - 1));
    memset(damp, 0, sizeof(real) * (T + 2 * M - 2));
    for (int t = 0; t < T; ++t)
        for (int m = 0; m < M; ++m) {
            const real ur = u[2 * (t + m)], ui = u[2 * (t + m) + 1];
            const real g = dy[2 * t] * ur + dy[2 * t + 1] * ui;     /* d loss / d (real coefficient of u[t+m]) */
            real e = w[m];
            dp[m] += g;
            for (int d = 1; d < GMP_DEGREE; ++d)
                for (int i = 0; i < M; ++i) {
                    const int k = M + ((d - 1) * M + i) * M + m;
                    const real a = amp[t + i + m];
                    e += w[k] * gmp_pow(a, d);
                    dp[k] += g * gmp_pow(a, d);
                    damp[t + i + m] += w[k] * g * (real)d * (d == 1 ? (real)1 : gmp_pow(a, d - 1));
                }
            du[2 * (t + m)] += dy[2 * t] * e;
            du[2 * (t + m) + 1] += dy[2 * t + 1] * e;
        }
    if (!dx) return;
    for (int k = 0; k < T + M - 1; ++k) {       /* amp[k + M-1] = |u[k]|; d|u|/du = u/|u| (0 at the origin, as torch.abs) */
        const real a = amp[k + M - 1];
        if (a > 0) { du[2 * k] += damp[k + M - 1] * u[2 * k] / a; du[2 * k + 1] += damp[k + M - 1] * u[2 * k + 1] / a; }
    }
    for (int s = 0; s < T; ++s) { dx[2 * s] = du[2 * (s + M - 1)]; dx[2 * s + 1] = du[2 * (s + M - 1) + 1]; }
}

/* ------------------------------------------------------------------------------------------ */
/* RVTDCNN: rvtdcnn.py:35-62.  Per sample t a 4 x 5 patch: window rows w = 0..3 hold the features [I,Q,a,a^2,a^3] (:41-46) of
 * samples t-3+w, the frame's own LAST three samples standing in front of it (:51-53: circular, not zero history);
 * Conv2d(1->3, k3, padding (1,0)) (:19-26) -> tanh -> flatten (channel, row, column) = 36 -> Linear(36->H) -> tanh ->
 * Linear(H->2) (:57-61).  Parameter order: Conv2d.weight (3,1,3,3), Conv2d.bias, fc_hid.weight (H,36), fc_hid.bias,
 * fc_out.weight (2,H), fc_out.bias. */
/* ------------------------------------------------------------------------------------------ */
#define RV_Z 36
static inline int rv_idx(int t, int w, int T) { int j = t - 3 + w; return ((j % T) + T) % T; }
static void rv_feat(const real* x, int s, real* f) {
    const real I = x[2 * s], Q = x[2 * s + 1];
    const real a2 = I * I + Q * Q, a = (real)sqrt((double)a2);
    f[0] = I; f[1] = Q; f[2] = a; f[3] = a2; f[4] = a * a * a;
}
/* forward of one sample; z (36) and hid (H) are kept for the backward */
static void rv_sample_fwd(int H, const real* p, int T, const real* x, int t, real in[4][5], real* z, real* hid, real* y) {
    const real* K = p; const real* kb = p + 27; const real* wh = p + 30; const real* bh = wh + 36 * H;
    const real* wo = bh + H; const real* bo = wo + 2 * H;
    for (int w = 0; w < 4; ++w) rv_feat(x, rv_idx(t, w, T), in[w]);
    for (int c = 0; c < 3; ++c)
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < 3; ++j) {
                real acc = kb[c];
                for (int dw = 0; dw < 3; ++dw) {
                    const int r = w + dw - 1;
                    if (r < 0 || r > 3) continue;                  /* zero padding of the window rows (padding=(1,0)) */
                    for (int dj = 0; dj < 3; ++dj) acc += K[(c * 3 + dw) * 3 + dj] * in[r][j + dj];
                }
                z[(c * 4 + w) * 3 + j] = tanhr(acc);
            }
    for (int u = 0; u < H; ++u) {
        real acc = bh[u];
        for (int k = 0; k < RV_Z; ++k) acc += wh[u * RV_Z + k] * z[k];
        hid[u] = tanhr(acc);
    }
    for (int c = 0; c < 2; ++c) {
        real acc = bo[c];
        for (int u = 0; u < H; ++u) acc += wo[c * H + u] * hid[u];
        y[c] = acc;
    }
}
static void rv_seq_fwd(int H, const real* p, int T, const real* x, real* y) {
    real in[4][5], z[RV_Z], hid[MAXH];
    for (int t = 0; t < T; ++t) rv_sample_fwd(H, p, T, x, t, in, z, hid, y + 2 * t);
}
/* dfeat: scratch of T x 5 reals (gradient w.r.t. the per-sample features, gathered over the <= 4 windows a sample sits in) */
static void rv_seq_bwd(int H, const real* p, int T, const real* x, const real* dy, real* dp, real* dx, real* dfeat) {
    const real* K = p; const real* wh = p + 30; const real* wo = wh + 36 * H + H;
    real* dK = dp; real* dkb = dp + 27; real* dwh = dp + 30; real* dbh = dwh + 36 * H; real* dwo = dbh + H; real* dbo = dwo + 2 * H;
    memset(dfeat, 0, sizeof(real) * 5 * T);
    for (int t = 0; t < T; ++t) {
        real in[4][5], z[RV_Z], hid[MAXH], yy[2], dz[RV_Z], din[4][5];
        rv_sample_fwd(H, p, T, x, t, in, z, hid, yy);
        const real d0 = dy[2 * t], d1 = dy[2 * t + 1];
        dbo[0] += d0; dbo[1] += d1;
        for (int k = 0; k < RV_Z; ++k) dz[k] = 0;
        for (int u = 0; u < H; ++u) {
            dwo[u] += d0 * hid[u]; dwo[H + u] += d1 * hid[u];
            const real dh = (wo[u] * d0 + wo[H + u] * d1) * ((real)1 - hid[u] * hid[u]);
            dbh[u] += dh;
            for (int k = 0; k < RV_Z; ++k) { dwh[u * RV_Z + k] += dh * z[k]; dz[k] += wh[u * RV_Z + k] * dh; }
        }
        memset(din, 0, sizeof(din));
        for (int c = 0; c < 3; ++c)
            for (int w = 0; w < 4; ++w)
                for (int j = 0; j < 3; ++j) {
                    const int k = (c * 4 + w) * 3 + j;
                    const real dc = dz[k] * ((real)1 - z[k] * z[k]);
                    dkb[c] += dc;
                    for (int dw = 0; dw < 3; ++dw) {
                        const int r = w + dw - 1;
                        if (r < 0 || r > 3) continue;
                        for (int dj = 0; dj < 3; ++dj) {
                            dK[(c * 3 + dw) * 3 + dj] += dc * in[r][j + dj];
                            din[r][j + dj] += K[(c * 3 + dw) * 3 + dj] * dc;
                        }
                    }
                }
        for (int w = 0; w < 4; ++w)
            for (int f = 0; f < 5; ++f) dfeat[5 * rv_idx(t, w, T) + f] += din[w][f];
    }
    if (!dx) return;
    for (int s = 0; s < T; ++s) {      /* features [I,Q,a,a^2,a^3]: da/dI = I/a, da^2/dI = 2I, da^3/dI = 3 a I */
        const real I = x[2 * s], Q = x[2 * s + 1];
        const real a = (real)sqrt((double)(I * I + Q * Q));
        const real* g = dfeat + 5 * s;
        const real ga = g[2] / a + (real)2 * g[3] + (real)3 * a * g[4];
        dx[2 * s] = g[0] + ga * I;
        dx[2 * s + 1] = g[1] + ga * Q;
    }
}

/* `--quant` on rvtdcnn (bits_w > 0): Conv2d -> INT_Conv2D (quant_layers.py:10-45: weight and activation quantisers; the weight scale
 * starts at 2 mean|w| / sqrt(Qp), init_step_size, and is rounded to a power of two like every scale), fc_hid / fc_out -> INT_Linear
 * (:48-85), fc_out with the 16-bit output quantiser in eval mode; the functional tanh calls stay float.  Parameter order: Conv2d.weight,
 * Conv2d.bias, its two scales; fc_hid.weight, .bias, three scales; fc_out.weight, .bias, three scales.  Scale gradients: exactly 0. */
typedef struct { int H, bw, ba, eval; int64_t oK, okb, oqc, owh, obh, oqh, owo, obo, oqo, P; } rvq_layout_t;
static void rvq_layout(const odpd_model_t* m, rvq_layout_t* g) {
    int64_t H = m->hidden, o = 0;
    g->H = (int)H; g->bw = m->bits_w; g->ba = m->bits_a; g->eval = m->flags & 1;
    g->oK = o; o += 27; g->okb = o; o += 3; g->oqc = o; o += 2;
    g->owh = o; o += 36 * H; g->obh = o; o += H; g->oqh = o; o += 3;
    g->owo = o; o += 2 * H; g->obo = o; o += 2; g->oqo = o; o += 3;
    g->P = o;
}
typedef struct { real inq[4][5], pin[4][5], z[RV_Z], zq[RV_Z], pz[RV_Z], hid[MAXH], hq[MAXH], ph[MAXH]; } rvq_sample_t;
static void rvq_sample_fwd(const rvq_layout_t* L, const real* p, int T, const real* x, int t, rvq_sample_t* s, real* y) {
    const int H = L->H;
    const real scw = q_pow2(p[L->oqc]), sca = q_pow2(p[L->oqc + 1]), shw = q_pow2(p[L->oqh]), sha = q_pow2(p[L->oqh + 1]);
    const real sow = q_pow2(p[L->oqo]), soa = q_pow2(p[L->oqo + 1]);
    for (int w = 0; w < 4; ++w) {
        real f[5];
        rv_feat(x, rv_idx(t, w, T), f);
        for (int j = 0; j < 5; ++j) s->inq[w][j] = q_apply(f[j], sca, L->ba, &s->pin[w][j]);
    }
    for (int c = 0; c < 3; ++c)
        for (int w = 0; w < 4; ++w)
            for (int j = 0; j < 3; ++j) {
                double acc = 0;
                for (int dw = 0; dw < 3; ++dw) {
                    const int r = w + dw - 1;
                    if (r < 0 || r > 3) continue;
                    for (int dj = 0; dj < 3; ++dj) acc += (double)q_apply(p[L->oK + (c * 3 + dw) * 3 + dj], scw, L->bw, NULL) * (double)s->inq[r][j + dj];
                }
                s->z[(c * 4 + w) * 3 + j] = tanhr((real)acc + p[L->okb + c]);
            }
    for (int k = 0; k < RV_Z; ++k) s->zq[k] = q_apply(s->z[k], sha, L->ba, &s->pz[k]);
    for (int u = 0; u < H; ++u) {
        double acc = 0;
        for (int k = 0; k < RV_Z; ++k) acc += (double)q_apply(p[L->owh + u * RV_Z + k], shw, L->bw, NULL) * (double)s->zq[k];
        s->hid[u] = tanhr((real)acc + p[L->obh + u]);
        s->hq[u] = q_apply(s->hid[u], soa, L->ba, &s->ph[u]);
    }
    for (int c = 0; c < 2; ++c) {
        double acc = 0;
        for (int u = 0; u < H; ++u) acc += (double)q_apply(p[L->owo + c * H + u], sow, L->bw, NULL) * (double)s->hq[u];
        y[c] = (real)acc + p[L->obo + c];
        if (L->eval) y[c] = q_apply(y[c], q_pow2(p[L->oqo + 2]), 16, NULL);
    }
}
static void rvq_seq(const rvq_layout_t* L, const real* p, int T, const real* x, real* y, const real* dy, real* dp, real* dx, real* dfeat) {
    const int H = L->H;
    const real scw = q_pow2(p[L->oqc]), shw = q_pow2(p[L->oqh]), sow = q_pow2(p[L->oqo]);
    rvq_sample_t s;
    if (!dy) { for (int t = 0; t < T; ++t) rvq_sample_fwd(L, p, T, x, t, &s, y + 2 * t); return; }
    memset(dfeat, 0, sizeof(real) * 5 * T);
    for (int t = 0; t < T; ++t) {
        real yy[2], dz[RV_Z], din[4][5];
        rvq_sample_fwd(L, p, T, x, t, &s, yy);
        const real d[2] = {dy[2 * t], dy[2 * t + 1]};
        dp[L->obo] += d[0]; dp[L->obo + 1] += d[1];
        for (int k = 0; k < RV_Z; ++k) dz[k] = 0;
        for (int u = 0; u < H; ++u) {
            real dhq = 0;
            for (int c = 0; c < 2; ++c) {
                real mk, wq = q_apply(p[L->owo + c * H + u], sow, L->bw, &mk);
                dp[L->owo + c * H + u] += d[c] * s.hq[u] * mk;
                dhq += wq * d[c];
            }
            const real dh = dhq * s.ph[u] * ((real)1 - s.hid[u] * s.hid[u]);
            dp[L->obh + u] += dh;
            for (int k = 0; k < RV_Z; ++k) {
                real mk, wq = q_apply(p[L->owh + u * RV_Z + k], shw, L->bw, &mk);
                dp[L->owh + u * RV_Z + k] += dh * s.zq[k] * mk;
                dz[k] += wq * dh;
            }
        }
        memset(din, 0, sizeof(din));
        for (int c = 0; c < 3; ++c)
            for (int w = 0; w < 4; ++w)
                for (int j = 0; j < 3; ++j) {
                    const int k = (c * 4 + w) * 3 + j;
                    const real dc = dz[k] * s.pz[k] * ((real)1 - s.z[k] * s.z[k]);
                    dp[L->okb + c] += dc;
                    for (int dw = 0; dw < 3; ++dw) {
                        const int r = w + dw - 1;
                        if (r < 0 || r > 3) continue;
                        for (int dj = 0; dj < 3; ++dj) {
                            real mk, wq = q_apply(p[L->oK + (c * 3 + dw) * 3 + dj], scw, L->bw, &mk);
                            dp[L->oK + (c * 3 + dw) * 3 + dj] += dc * s.inq[r][j + dj] * mk;
                            din[r][j + dj] += wq * dc;
                        }
                    }
                }
        for (int w = 0; w < 4; ++w)
            for (int f = 0; f < 5; ++f) dfeat[5 * rv_idx(t, w, T) + f] += din[w][f] * s.pin[w][f];
    }
    if (!dx) return;
    for (int sx = 0; sx < T; ++sx) {
        const real I = x[2 * sx], Q = x[2 * sx + 1];
        const real a = (real)sqrt((double)(I * I + Q * Q));
        const real* g = dfeat + 5 * sx;
        const real ga = g[2] / a + (real)2 * g[3] + (real)3 * a * g[4];
        dx[2 * sx] = g[0] + ga * I;
        dx[2 * sx + 1] = g[1] + ga * Q;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* dispatch                                                                                     */
/* ------------------------------------------------------------------------------------------ */
static int is_gru_family(int bb) { return bb == ODPD_GRU || bb == ODPD_DGRU || bb == ODPD_QGRU || bb == ODPD_QGRU_AMP1; }
static int is_lstm_family(int bb) { return bb == ODPD_LSTM || bb == ODPD_VDLSTM; }
static int is_delta_family(int bb) { return bb == ODPD_DELTAGRU || bb == ODPD_TRES_DELTAGRU; }

/* generic per-sequence runner: fwd only (dy == NULL) or fwd + bwd */
static void seq_run(const odpd_model_t* m, int T, const real* params, const real* x, real* y, const real* dy, real* dp, real* dx,
                    double* stats, void* scratch) {
    int bb = m->backbone;
    if (is_gru_family(bb)) {
        gru_params_t g; gru_layout(m, params, &g);
        gru_step_t* S = (gru_step_t*)scratch;
        gru_seq_fwd(m, &g, T, x, y, dy ? S : NULL);
        if (dy) gru_seq_bwd(m, &g, T, x, dy, S, dp, dx);
    } else if (is_lstm_family(bb)) {
        lstm_layout_t L; lstm_layout(m, &L);
        lstm_step_t* S = (lstm_step_t*)scratch;
        lstm_seq_fwd(&L, params, T, x, y, dy ? S : NULL);
        if (dy) lstm_seq_bwd(&L, params, T, x, dy, S, dp, dx);
    } else if (is_delta_family(bb)) {
        delta_layout_t L; delta_layout(m, &L);
        delta_step_t* S = (delta_step_t*)scratch;
        delta_seq_fwd(m, &L, params, T, x, y, dy ? S : NULL, stats);
        if (dy) delta_seq_bwd(m, &L, params, T, x, dy, S, dp, dx);
    } else if (bb == ODPD_TCNN) {
        tcnn_layout_t L; tcnn_layout(m, &L);
        real* feat = (real*)scratch; real* pre = feat + (size_t)T * 6;
        real* g1 = pre + (size_t)5 * T * L.C; real* g2 = g1 + (size_t)T * L.C;
        tcnn_seq_fwd(&L, params, T, x, y, feat, pre);
        if (dy) tcnn_seq_bwd(&L, params, T, x, dy, feat, pre, dp, dx, g1, g2);
    } else if (bb == ODPD_PGJANET && m->bits_w > 0) {
        pgq_layout_t L; pgq_layout(m, &L);
        pgq_step_t* S = (pgq_step_t*)scratch;
        pgq_seq_fwd(&L, params, T, x, y, dy ? S : NULL);
        if (dy) pgq_seq_bwd(&L, params, T, x, dy, S, dp, dx);
    } else if (bb == ODPD_PGJANET) {
        pgj_layout_t L; pgj_layout(m, &L);
        pgj_step_t* S = (pgj_step_t*)scratch;
        pgj_seq_fwd(&L, params, T, x, y, dy ? S : NULL);
        if (dy) pgj_seq_bwd(&L, params, T, x, dy, S, dp, dx);
    } else if (bb == ODPD_GMP) {
        const int M = m->hidden;
        real* u = (real*)scratch; real* amp = u + 2 * (T + M - 1);
        real* du = amp + (T + 2 * M - 2); real* damp = du + 2 * (T + M - 1);
        if (dy) gmp_seq_bwd(M, params, T, x, dy, dp, dx, u, amp, du, damp);
        else gmp_seq_fwd(M, params, T, x, y, u, amp);
    } else if (bb == ODPD_DVRJANET) {
        dvr_layout_t L; dvr_layout(m, &L);
        dvr_step_t* S = (dvr_step_t*)scratch;
        dvr_seq_fwd(&L, params, T, x, y, dy ? S : NULL);
        if (dy) dvr_seq_bwd(&L, params, T, x, dy, S, dp, dx);
    } else if (bb == ODPD_BOJANET) {
        boj_layout_t L; boj_layout(m, &L);
        boj_step_t* S = (boj_step_t*)scratch;
        boj_seq_fwd(&L, params, T, x, y, dy ? S : NULL);
        if (dy) boj_seq_bwd(&L, params, T, x, dy, S, dp, dx);
    } else if (bb == ODPD_APNRRU) {
        apn_layout_t L; apn_layout(m, &L);
        apn_step_t* S = (apn_step_t*)scratch;
        apn_seq_fwd(&L, params, T, x, y, dy ? S : NULL);
        if (dy) apn_seq_bwd(&L, params, T, x, dy, S, dp, dx);
    } else if (bb == ODPD_MCLDNN) {
        mcl_layout_t L; mcl_layout(m, &L);
        mcl_step_t* S = (mcl_step_t*)scratch;
        real* work = (real*)(S + T);
        mcl_seq_fwd(&L, params, T, x, y, dy ? S : NULL, work);
        if (dy) mcl_seq_bwd(&L, params, T, x, dy, S, dp, dx, work);
    } else if (bb == ODPD_DELTAJANET) {
        dj_layout_t L; dj_layout(m, &L);
        dj_step_t* S = (dj_step_t*)scratch;
        dj_seq_fwd(&L, params, T, x, y, dy ? S : NULL, stats);
        if (dy) dj_seq_bwd(&L, params, T, x, dy, S, dp, dx);
    } else if (bb == ODPD_NEURALTX) {
        ntx_layout_t L; ntx_layout(m, &L);
        real* feat = (real*)scratch; real* pre = feat + (size_t)T * 4;
        real* g1 = pre + (size_t)5 * T * L.C; real* g2 = g1 + (size_t)T * L.C; real* dfq = g2 + (size_t)T * L.C;
        ntx_seq_fwd(&L, params, T, x, y, feat, pre);
        if (dy) ntx_seq_bwd(&L, params, T, x, dy, feat, pre, dp, dx, g1, g2, dfq);
    } else if (bb == ODPD_RVTDCNN && m->bits_w > 0) {
        rvq_layout_t L; rvq_layout(m, &L);
        rvq_seq(&L, params, T, x, y, dy, dp, dx, (real*)scratch);
    } else if (bb == ODPD_RVTDCNN) {
        if (dy) rv_seq_bwd(m->hidden, params, T, x, dy, dp, dx, (real*)scratch);
        else rv_seq_fwd(m->hidden, params, T, x, y);
    }
}
static size_t seq_scratch_bytes(const odpd_model_t* m, int T) {
    int bb = m->backbone;
    if (is_gru_family(bb)) return sizeof(gru_step_t) * T;
    if (is_lstm_family(bb)) return sizeof(lstm_step_t) * T;
    if (is_delta_family(bb)) return sizeof(delta_step_t) * T;
    if (bb == ODPD_TCNN) return sizeof(real) * ((size_t)T * 6 + (size_t)7 * T * m->hidden);
    if (bb == ODPD_PGJANET) return (m->bits_w > 0 ? sizeof(pgq_step_t) : sizeof(pgj_step_t)) * T;
    if (bb == ODPD_GMP) return sizeof(real) * (size_t)(6 * (T + 2 * m->hidden));
    if (bb == ODPD_DVRJANET) return sizeof(dvr_step_t) * T;
    if (bb == ODPD_BOJANET) return T >= BOJ_M - 1 ? sizeof(boj_step_t) * T : 0;   /* bojanet.py:72-77 cannot frame fewer than 15 samples */
    if (bb == ODPD_APNRRU) return T >= APN_M - 1 ? sizeof(apn_step_t) * T : 0;   /* apnrru.py:68-72 cannot frame fewer than 15 samples */
    if (bb == ODPD_MCLDNN)     /* the circular window needs its four samples (mcldnn.py:115-118) */
        return T >= 4 ? sizeof(mcl_step_t) * T + sizeof(real) * (size_t)(2 * (10 * m->hidden * 5 + 5 * m->hidden) + 5 * T) : 0;
    if (bb == ODPD_DELTAJANET) return sizeof(dj_step_t) * T;
    if (bb == ODPD_NEURALTX) return sizeof(real) * ((size_t)T * 6 + (size_t)7 * T * m->hidden);
    if (bb == ODPD_RVTDCNN) return T >= 3 ? sizeof(real) * (size_t)(5 * T) : 0;   /* the circular window needs 3 samples */
    return 0;
}

int oracle_backbone_fwd(const odpd_model_t* m, int B, int T, const real* params, const real* x, real* y, double* stats) {
    if (!m || !params || !x || !y || B <= 0 || T <= 0 || m->hidden > MAXH || m->hidden <= 0) return ODPD_EINVAL;
    if (oracle_param_count(m) < 0 || !seq_scratch_bytes(m, T)) return ODPD_EUNSUPPORTED;
    double st[4] = {0, 0, 0, 0};
#pragma omp parallel
    {
        void* scratch = malloc(seq_scratch_bytes(m, T));
        double lst[4] = {0, 0, 0, 0};
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b)
            seq_run(m, T, params, x + (int64_t)b * T * 2, y + (int64_t)b * T * 2, NULL, NULL, NULL, lst, scratch);
#pragma omp critical
        for (int i = 0; i < 4; ++i) st[i] += lst[i];
        free(scratch);
    }
    if (stats) for (int i = 0; i < 4; ++i) stats[i] += st[i];
    return 0;
}

/* dparams (P) is OVERWRITTEN with the sum over the batch; dx nullable */
int oracle_backbone_bwd(const odpd_model_t* m, int B, int T, const real* params, const real* x, const real* dy,
                        real* dparams, real* dx) {
    if (!m || !params || !x || !dy || !dparams || B <= 0 || T <= 0 || m->hidden > MAXH || m->hidden <= 0) return ODPD_EINVAL;
    int64_t P = oracle_param_count(m);
    if (P < 0 || !seq_scratch_bytes(m, T)) return ODPD_EUNSUPPORTED;
    memset(dparams, 0, sizeof(real) * P);
#pragma omp parallel
    {
        real* dp = (real*)calloc(P, sizeof(real));
        void* scratch = malloc(seq_scratch_bytes(m, T));
        real* ytmp = (real*)malloc(sizeof(real) * 2 * T);
        double lst[4] = {0, 0, 0, 0};
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b)
            seq_run(m, T, params, x + (int64_t)b * T * 2, ytmp, dy + (int64_t)b * T * 2, dp, dx ? dx + (int64_t)b * T * 2 : NULL,
                    lst, scratch);
#pragma omp critical
        for (int64_t i = 0; i < P; ++i) dparams[i] += dp[i];
        free(dp); free(scratch); free(ytmp);
    }
    return 0;
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
/* number of OpenMP threads the batch loops use from now on (bench.py: 1-thread and all-core baselines) */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
