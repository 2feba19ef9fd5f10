// Micro-benchmark 3 (r05): v_mfma_f32_16x16x32_bf16 issue patterns for the bf16x3 kernels (gru_s16x.hip) —
// dependent chains on one accumulator, chains interleaved over 2 / 3 / 6 accumulators, VALU work between the MFMAs of a
// chain (same wave) and VALU phases after a chain (the other wave of the SIMD fills in), at 1 and 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16.hip -o tools/ubench/mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(C, A, B) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)
#define VOP(j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[(j) & 7]) : "v"(x), "v"(y));
#define BAR __builtin_amdgcn_sched_barrier(0);

// MODE: 0 = NACC accumulators round-robin, 6 MFMAs per accumulator per group, NV VALU ops after EVERY MFMA (same wave)
//       1 = the same MFMAs, the NV * (6 NACC) VALU ops in one phase AFTER the group
template <int NACC, int NV, int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    f32x4 c[NACC];
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, threadIdx.x, 8};
    float v[8], x = seed + threadIdx.x, y = seed * 0.5f;
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{seed, 0, 0, 0};
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                MFMA(c[t], a, b);
                BAR
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < NV; ++j) VOP(j)
                    BAR
                }
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < NV * 6 * NACC; ++j) VOP(j)
            BAR
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int wps, int nmfma, int nvalu) {
    float* d; hipMalloc(&d, 1 << 26);
    int blocks = 256, threads = 256 * wps, iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters);     // SIMD cycles per group per wave-slot
    printf("%-44s waves/SIMD %d: %8.1f cycles per group of %3d MFMA + %4d VALU  = %.1f per MFMA\n", name, wps, cyc, nmfma, nvalu, cyc / nmfma);
    hipFree(d);
}
#define RUN(NACC, NV, MODE) for (int w : {1, 2}) run("NACC=" #NACC " NV=" #NV " MODE=" #MODE, k<NACC, NV, MODE>, w, 6 * NACC, NV * 6 * NACC);
int main() {
    RUN(1, 0, 0) RUN(2, 0, 0) RUN(3, 0, 0) RUN(6, 0, 0)
    RUN(1, 1, 0) RUN(1, 4, 0) RUN(2, 4, 0) RUN(6, 4, 0) RUN(6, 6, 0)
    RUN(1, 4, 1) RUN(2, 4, 1) RUN(6, 4, 1) RUN(6, 6, 1) RUN(1, 6, 1)
    return 0;
}
