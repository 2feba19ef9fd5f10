// optim.hip — loss, deterministic gradient reduction, and fused clip_grad_norm_ + AdamW.
//   loss      : nn.MSELoss()/nn.L1Loss() mean reduction + backward   (reference project.py:262-272)
//   clip+step : nn.utils.clip_grad_norm_ (modules/train_funcs.py:41-42) followed by
//               torch.optim.AdamW single-tensor update (project.py:283; torch/optim/adam.py
//               _single_tensor_adam: lerp_, addcmul_, bias corrections in double, addcdiv_).
#include "odpd_host.h"
#include "odpd_xchg.h"

namespace odpd {

constexpr int kLossBlocks = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
    // deterministic: wave shuffle tree, then lane 0 of wave 0 sums the per-wave values in order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) t += sh[i];
    __syncthreads();
    return t;  // valid on thread 0
}

__global__ __launch_bounds__(256) void loss_kernel(int kind, int64_t n, float inv_count, const float* __restrict__ y,
                                                   const float* __restrict__ t, float* __restrict__ dy,
                                                   float* __restrict__ ws) {
    __shared__ float sh[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    float4* d4 = reinterpret_cast<float4*>(dy);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = y4[i], b = t4[i], g;
        float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        if (kind == ODPD_LOSS_L2) {
            acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            const float s = 2.0f * inv_count;
            g = make_float4(d0 * s, d1 * s, d2 * s, d3 * s);
        } else {
            acc += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
            auto sg = [inv_count](float d) { return d > 0.f ? inv_count : (d < 0.f ? -inv_count : 0.f); };
            g = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
        }
        if (dy) d4[i] = g;
    }
    // tail (n not a multiple of 4; n = B*T*2 is even, so at most 2 elements)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float d = y[i] - t[i];
        if (kind == ODPD_LOSS_L2) { acc += d * d; if (dy) dy[i] = 2.0f * inv_count * d; }
        else { acc += fabsf(d); if (dy) dy[i] = d > 0.f ? inv_count : (d < 0.f ? -inv_count : 0.f); }
    }
    float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) ws[blockIdx.x] = tot;
}
// the same for small batches (the reference's 64 x 50 frames: 6 400 elements) as ONE launch of one 1024-thread workgroup: per-thread sums over the
// grid-stride elements, one block sum, the mean written directly
__global__ __launch_bounds__(1024) void loss_single_kernel(int kind, int64_t n, float inv_count, const float* __restrict__ y,
                                                           const float* __restrict__ t, float* __restrict__ dy, float* __restrict__ out) {
    __shared__ float sh[16];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const float4* t4 = reinterpret_cast<const float4*>(t);
    float4* d4 = reinterpret_cast<float4*>(dy);
    for (int64_t i = threadIdx.x; i < n4; i += blockDim.x) {
        float4 a = y4[i], b = t4[i], g;
        float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        if (kind == ODPD_LOSS_L2) {
            acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            const float s = 2.0f * inv_count;
            g = make_float4(d0 * s, d1 * s, d2 * s, d3 * s);
        } else {
            acc += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
            auto sg = [inv_count](float d) { return d > 0.f ? inv_count : (d < 0.f ? -inv_count : 0.f); };
            g = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
        }
        if (dy) d4[i] = g;
    }
    if (threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float d = y[i] - t[i];
        if (kind == ODPD_LOSS_L2) { acc += d * d; if (dy) dy[i] = 2.0f * inv_count * d; }
        else { acc += fabsf(d); if (dy) dy[i] = d > 0.f ? inv_count : (d < 0.f ? -inv_count : 0.f); }
    }
    float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = tot * inv_count;
}
__global__ __launch_bounds__(256) void loss_final_kernel(int nblk, float inv_count, const float* __restrict__ ws,
                                                         float* __restrict__ out) {
    __shared__ float sh[4];
    float v = threadIdx.x < nblk ? ws[threadIdx.x] : 0.f;
    float tot = block_sum(v, sh);
    if (threadIdx.x == 0) out[0] = tot * inv_count;
}

// grad[c] = sum_r partials[r][c]; fixed summation order -> bit-repeatable.
// 64 columns per block; 16 waves split the rows (8 independent loads in flight per lane), then a
// fixed-order combine through LDS.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(int64_t rows, int64_t cols, const float* __restrict__ part,
                                                               float* __restrict__ grad, int accumulate) {
    __shared__ float sh[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (c < cols) {
        int64_t r = wave;
        for (; r + 7 * 16 < rows; r += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(r + u * 16) * cols + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; r < rows; r += 16) acc += part[r * cols + c];
    }
    sh[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols) {
        float v = sh[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) v += sh[w][lane];
        grad[c] = accumulate ? grad[c] + v : v;
    }
}

// single workgroup: P is ~1e3 (0.5-3 k parameters); everything stays in one CU.
// kind (enum odpd_optimizer): AdamW / Adam share the first branch (Adam = no decoupled decay);  SGD with momentum: buf = g at the first
// step, then buf = momentum buf + g, p -= lr buf (torch/optim/sgd.py, dampening 0);  RMSprop: sq = alpha sq + (1 - alpha) g^2,
// p -= lr g / (sqrt(sq) + eps) (torch/optim/rmsprop.py, no momentum, not centered).  Multiplies and adds are kept apart where torch
// issues them as separate ATen ops.
__device__ __forceinline__ void clip_optim_block(int kind, int64_t P, float* __restrict__ p, float* __restrict__ g,
                                                 float* __restrict__ m, float* __restrict__ v, float step_size,
                                                 float bc2_sqrt, float decay, float w1, float b2, float w2,
                                                 float eps, float max_norm, float* __restrict__ norm_out,
                                                 float* __restrict__ loss_out, float inv_count,
                                                 const unsigned char* __restrict__ skip, int first_step) {
    __shared__ float sh[16];
    __shared__ float coef_s;
    if (loss_out && threadIdx.x == 0) loss_out[0] = g[P] * inv_count;   // column P of the reduced row = loss partial sum
    float acc = 0.f;
    // skip[i] != 0: a parameter whose .grad is None in the reference — outside the norm, untouched by the update
    for (int64_t i = threadIdx.x; i < P; i += blockDim.x) acc += (skip && skip[i]) ? 0.f : g[i] * g[i];
    float tot = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        float nrm = sqrtf(tot);
        if (norm_out) norm_out[0] = nrm;
        float coef = 1.0f;
        if (max_norm > 0.f) coef = fminf(max_norm / (nrm + 1e-6f), 1.0f);
        coef_s = coef;
    }
    __syncthreads();
    const float coef = coef_s;
    for (int64_t i = threadIdx.x; i < P; i += blockDim.x) {
        if (skip && skip[i]) continue;
        float gi = g[i];
        if (max_norm > 0.f) { gi *= coef; g[i] = gi; }
        if (kind == ODPD_OPT_SGD) {             // step_size = lr, b2 = momentum
            const float bi = first_step ? gi : __fadd_rn(__fmul_rn(m[i], b2), gi);
            m[i] = bi;
            p[i] = __fadd_rn(p[i], __fmul_rn(-step_size, bi));
        } else if (kind == ODPD_OPT_RMSPROP) {  // step_size = lr, b2 = alpha, w2 = 1 - alpha
            const float vi = __fadd_rn(__fmul_rn(v[i], b2), __fmul_rn(w2, __fmul_rn(gi, gi)));
            v[i] = vi;
            p[i] = __fadd_rn(p[i], __fmul_rn(-step_size, __fdiv_rn(gi, __fadd_rn(__fsqrt_rn(vi), eps))));
        } else {
            float pi = p[i] * decay;
            float mi = m[i] + (gi - m[i]) * w1;
            float vi = v[i] * b2 + w2 * gi * gi;
            float denom = sqrtf(vi) / bc2_sqrt + eps;
            pi -= step_size * (mi / denom);
            p[i] = pi; m[i] = mi; v[i] = vi;
        }
    }
}
__global__ __launch_bounds__(1024) void clip_optim_kernel(int kind, int64_t P, float* __restrict__ p, float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, float step_size,
                                                          float bc2_sqrt, float decay, float w1, float b2, float w2,
                                                          float eps, float max_norm, float* __restrict__ norm_out,
                                                          float* __restrict__ loss_out, float inv_count,
                                                          const unsigned char* __restrict__ skip, int first_step, int n_xchg,
                                                          XchgDev xd) {
    // data parallel over a one-shot communicator: the step's collective is this kernel's prologue — g[0 .. P+4) becomes the sum over
    // the ranks (gradient + loss partial sum), identical bits on every rank
    if (n_xchg > 0) xchg_allreduce_block(xd, g, n_xchg);
    clip_optim_block(kind, P, p, g, m, v, step_size, bc2_sqrt, decay, w1, b2, w2, eps, max_norm, norm_out, loss_out, inv_count, skip, first_step);
}
// ---- lockstep sweeps (odpd_train_epoch_sweep): the same two kernels for K runs at once, run k = the workgroup(s) blockIdx.x / per-run count ----
__global__ __launch_bounds__(1024) void reduce_partials_sweep_kernel(const SweepRun* __restrict__ runs, int cb, int64_t rows, int64_t cols) {
    __shared__ float sh[16][64];
    const SweepRun r = runs[blockIdx.x / cb];
    const float* __restrict__ part = r.partials;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)(blockIdx.x % cb) * 64 + lane;
    float acc = 0.f;
    if (c < cols) {      // (the loop of reduce_partials_kernel, instruction for instruction: the sums must not differ by a bit)
        int64_t rr = wave;
        for (; rr + 7 * 16 < rows; rr += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(rr + u * 16) * cols + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; rr < rows; rr += 16) acc += part[rr * cols + c];
    }
    sh[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols) {
        float v = sh[0][lane];
#pragma unroll
        for (int w = 1; w < 16; ++w) v += sh[w][lane];
        r.grad[c] = v;
    }
}
__global__ __launch_bounds__(1024) void clip_adamw_sweep_kernel(const SweepRun* __restrict__ runs, int64_t P, const float* __restrict__ step_sizes,
                                                                float bc2_sqrt, float w1, float b2, float w2, float eps, float max_norm,
                                                                int64_t loss_index, float inv_count) {
    const SweepRun r = runs[blockIdx.x];
    clip_optim_block((int)ODPD_OPT_ADAMW, P, r.params, r.grad, r.state1, r.state2, step_sizes[blockIdx.x], bc2_sqrt, r.decay, w1, b2, w2, eps, max_norm,
                     nullptr, r.losses + loss_index, inv_count, nullptr, 0);
}

}  // namespace odpd

using namespace odpd;

extern "C" int odpd_loss_fwd_bwd(void* stream, int kind, int64_t n, int64_t count, const float* y, const float* target,
                                 float* dy, float* loss_out) {
    if (!y || !target || !loss_out || n <= 0 || count <= 0 || (kind != ODPD_LOSS_L2 && kind != ODPD_LOSS_L1))
        return ODPD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int64_t want = (n / 4 + 255) / 256;
    int nblk = (int)(want < 1 ? 1 : (want > kLossBlocks ? kLossBlocks : want));
    float inv = (float)(1.0 / (double)count);
    if (n <= 32768) {       // one workgroup, one launch
        hipLaunchKernelGGL(loss_single_kernel, dim3(1), dim3(1024), 0, st, kind, n, inv, y, target, dy, loss_out);
        return (int)hipGetLastError();
    }
    // loss_out[1..256] is scratch for the per-block sums (see header)
    hipLaunchKernelGGL(loss_kernel, dim3(nblk), dim3(256), 0, st, kind, n, inv, y, target, dy, loss_out + 1);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, nblk, inv, loss_out + 1, loss_out);
    return (int)hipGetLastError();
}

extern "C" int odpd_reduce_partials(void* stream, int64_t rows, int64_t P, const float* partials, float* grad,
                                    int accumulate) {
    if (!partials || !grad || rows <= 0 || P < 0) return ODPD_EINVAL;      // P = 0: loss rows only (odpd_frozen_loss_dx)
    const int64_t cols = P + kLossCols;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, rows,
                       cols, partials, grad, accumulate);
    return (int)hipGetLastError();
}

int odpd::launch_clip_adamw(hipStream_t st, int64_t P, float* params, float* grad, float* exp_avg, float* exp_avg_sq,
                            int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay,
                            double max_norm, float* norm_out, float* loss_out, float inv_count, const unsigned char* skip, const XchgDev* xchg) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || P <= 0 || step <= 0) return ODPD_EINVAL;
    // bias corrections in double like torch/optim/adam.py (_single_tensor_adam), rounded to fp32 once
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2);
    const float decay = (float)(1.0 - lr * weight_decay);
    const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2);
    hipLaunchKernelGGL(clip_optim_kernel, dim3(1), dim3(1024), 0, st, (int)ODPD_OPT_ADAMW, P, params, grad, exp_avg, exp_avg_sq, step_size,
                       bc2s, decay, w1, (float)beta2, w2, (float)eps, (float)max_norm, norm_out, loss_out, inv_count, skip, 0,
                       xchg ? (int)(P + kLossCols) : 0, xchg ? *xchg : XchgDev{});
    return (int)hipGetLastError();
}

int odpd::launch_reduce_sweep(hipStream_t st, const SweepRun* runs, int K, int64_t rows, int64_t P) {
    const int64_t cols = P + kLossCols;
    const int cb = (int)((cols + 63) / 64);
    hipLaunchKernelGGL(reduce_partials_sweep_kernel, dim3((unsigned)(cb * K)), dim3(1024), 0, st, runs, cb, rows, cols);
    return (int)hipGetLastError();
}
// (the per-run step size lr_k / (1 - beta1^step) comes precomputed — in double, rounded once, as launch_clip_adamw forms it)
int odpd::launch_clip_adamw_sweep(hipStream_t st, const SweepRun* runs, int K, int64_t P, const float* step_sizes, int64_t step, int64_t loss_index,
                                  double beta1, double beta2, double eps, double max_norm, float inv_count) {
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(clip_adamw_sweep_kernel, dim3((unsigned)K), dim3(1024), 0, st, runs, P, step_sizes, (float)sqrt(bc2), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, (float)max_norm, loss_index, inv_count);
    return (int)hipGetLastError();
}

// the optimisers of project.py:274-297 with the hyper-parameters the reference constructs them with
int odpd::launch_clip_optim(hipStream_t st, int kind, int64_t P, float* params, float* grad, float* state1, float* state2, int64_t step,
                            double lr, double max_norm, float* norm_out, float* loss_out, float inv_count, const unsigned char* skip,
                            const XchgDev* xchg) {
    switch (kind) {
    case ODPD_OPT_ADAMW: return launch_clip_adamw(st, P, params, grad, state1, state2, step, lr, 0.9, 0.999, 1e-8, 0.01, max_norm, norm_out, loss_out, inv_count, skip, xchg);
    case ODPD_OPT_ADAM: return launch_clip_adamw(st, P, params, grad, state1, state2, step, lr, 0.9, 0.999, 1e-8, 0.0, max_norm, norm_out, loss_out, inv_count, skip, xchg);
    case ODPD_OPT_SGD: case ODPD_OPT_RMSPROP: break;
    default: return ODPD_EINVAL;
    }
    if (!params || !grad || !state1 || !state2 || P <= 0 || step <= 0) return ODPD_EINVAL;
    const bool sgd = kind == ODPD_OPT_SGD;
    const float b2 = sgd ? 0.9f : 0.99f, w2 = sgd ? 0.0f : (float)(1.0 - 0.99);
    hipLaunchKernelGGL(clip_optim_kernel, dim3(1), dim3(1024), 0, st, kind, P, params, grad, state1, state2, (float)lr, 1.0f, 1.0f, 0.0f, b2, w2,
                       1e-8f, (float)max_norm, norm_out, loss_out, inv_count, skip, step == 1 ? 1 : 0, xchg ? (int)(P + kLossCols) : 0,
                       xchg ? *xchg : XchgDev{});
    return (int)hipGetLastError();
}
extern "C" int odpd_clip_optim_step(void* stream, int kind, int64_t P, float* params, float* grad, float* state1, float* state2,
                                    int64_t step, double lr, double max_norm, float* norm_out, const unsigned char* skip) {
    return launch_clip_optim((hipStream_t)stream, kind, P, params, grad, state1, state2, step, lr, max_norm, norm_out, nullptr, 0.0f, skip);
}

extern "C" int odpd_clip_adamw_step(void* stream, int64_t P, float* params, float* grad, float* exp_avg,
                                    float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double eps,
                                    double weight_decay, double max_norm, float* norm_out) {
    return launch_clip_adamw((hipStream_t)stream, P, params, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps,
                             weight_decay, max_norm, norm_out, nullptr, 0.0f, nullptr);
}

extern "C" int odpd_clip_adamw_step_masked(void* stream, int64_t P, float* params, float* grad, float* exp_avg,
                                           float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double eps,
                                           double weight_decay, double max_norm, float* norm_out, const unsigned char* skip) {
    return launch_clip_adamw((hipStream_t)stream, P, params, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps,
                             weight_decay, max_norm, norm_out, nullptr, 0.0f, skip);
}
