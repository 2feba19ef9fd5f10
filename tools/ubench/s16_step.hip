// Micro-benchmark 3: anatomy of one S16 forward step (18 exact-fp32 MFMAs in 4 accumulator chains + the GRU
// element-wise stage) — which part of the step time is MFMA issue, MFMA->VALU->MFMA latency, and VALU work.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench/s16_step.hip -o tools/ubench/s16_step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// MODE 0: MFMAs only, operands loop-invariant
// MODE 1: B operands of the 12 recurrent MFMAs = h, h = cheap VALU function of the accumulators (loop-carried)
// MODE 2: MODE 1 with the full element-wise stage (2 sigmoid + tanh + blend)
// MODE 3: element-wise stage only (no MFMA; accumulators = h-dependent VALU)
template <int MODE>
__global__ __launch_bounds__(256) void k_step(float* out, int iters, float seed) {
    float w[18];
    for (int i = 0; i < 18; ++i) w[i] = seed * 0.01f * (i + 1) + threadIdx.x * 1e-4f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, bias = {0.1f, 0.2f, 0.3f, 0.4f};
    f32x4 h = {seed, seed * 0.5f, seed * 0.25f, seed * 0.125f};
    float f0 = seed * 0.3f, f1 = seed * 0.7f;
    for (int it = 0; it < iters; ++it) {
        f32x4 ar = zero, az = zero, an = zero, ah = bias;
        if (MODE != 3) {
            ar = mfma4(w[0], f0, ar); az = mfma4(w[1], f0, az); an = mfma4(w[2], f0, an);
            ar = mfma4(w[3], f1, ar); az = mfma4(w[4], f1, az); an = mfma4(w[5], f1, an);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float b = MODE == 0 ? f0 : h[c];
                ar = mfma4(w[6 + c], b, ar); az = mfma4(w[10 + c], b, az); ah = mfma4(w[14 + c], b, ah);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { ar[i] = h[i] * w[i]; az[i] = h[i] * w[4 + i]; an[i] = h[i] * w[8 + i]; ah[i] = h[i] + w[12 + i]; }
        }
        if (MODE == 0) {
            h += ar + az + an + ah;   // outside the dependency chain of the next iteration's MFMAs
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = __builtin_fmaf(ar[i], az[i], an[i] * ah[i]) * 1e-3f;
        } else {
            f32x4 r, z, n;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ar[i]));
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(az[i]));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float p = __builtin_fmaf(r[i], ah[i], an[i]);
                n[i] = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.885390f * p)), -2.0f, 1.0f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = __builtin_fmaf(z[i], h[i] - n[i], n[i]);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h[0] + h[1] + h[2] + h[3];
}

template <typename K>
void run(const char* name, K k, int wps) {
    float* d; hipMalloc(&d, 1 << 26);
    const int blocks = 256 * wps, iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s waves/SIMD %d : %8.1f ns per step per wave, %8.1f ns per step per SIMD-slot\n", name, wps, ms * 1e6 / iters,
           ms * 1e6 / iters / wps);
    hipFree(d);
}
int main() {
    for (int wps = 1; wps <= 4; ++wps) {
        if (wps == 3) continue;
        run("0: 18 MFMA, loop-invariant operands", k_step<0>, wps);
        run("1: 18 MFMA, h = cheap VALU(acc) fed back", k_step<1>, wps);
        run("2: 18 MFMA + full element-wise, fed back", k_step<2>, wps);
        run("3: element-wise only", k_step<3>, wps);
    }
    return 0;
}
