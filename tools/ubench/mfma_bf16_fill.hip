// Micro-benchmark 4 (r05): how many VALU instructions of which kind hide in the shadow of a bf16 MFMA?
// NACC = 6 independent accumulators round-robin, NV fillers after every MFMA (same wave), 1 / 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BAR __builtin_amdgcn_sched_barrier(0);
template <int OP> __device__ __forceinline__ void vop(float (&v)[8], int j, float x, float y) {
    if (OP == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j & 7]) : "v"(x), "v"(y));
    if (OP == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(v[j & 7]) : "v"(x));
    if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[j & 7]) : "v"(x), "v"(y));
    if (OP == 3) asm volatile("v_and_b32 %0, %1, %2" : "=v"(v[j & 7]) : "v"(x), "v"(y));
    if (OP == 4) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[j & 7]) : "v"(x), "v"(y));
    if (OP == 5) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[j & 7]) : "v"(x));
}
template <int BIG, int NV, int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    constexpr int NACC = BIG ? 3 : 6;
    f32x4 c[6]; f32x16 C[3];
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, threadIdx.x, 8};
    float v[8], x = seed + threadIdx.x, y = seed * 0.5f;
    for (int i = 0; i < 6; ++i) c[i] = f32x4{seed, 0, 0, 0};
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 16; ++e) C[i][e] = seed;
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 6; ++p) {
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if (BIG) C[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), C[t], 0, 0, 0);
                else c[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c[t], 0, 0, 0);
                BAR
#pragma unroll
                for (int j = 0; j < NV; ++j) vop<OP>(v, j, x, y);
                BAR
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 6; ++i) s += c[i][0] + c[i][3];
    for (int i = 0; i < 3; ++i) s += C[i][0] + C[i][15];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
double run(K kern, int wps, int nmfma) {
    float* d; (void)hipMalloc(&d, 1 << 26);
    int iters = 3000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 0, 0, d, 100, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(d);
    return ms * 1e-3 * 2.4e9 / ((double)wps * iters * nmfma);
}
#define ROW(BIG, OP, NAME) { printf("%-34s", NAME); for (int w : {1, 2}) { printf(" | w%d:", w); \
    printf(" %5.1f", run(k<BIG, 0, OP>, w, BIG ? 18 : 36)); printf(" %5.1f", run(k<BIG, 1, OP>, w, BIG ? 18 : 36)); \
    printf(" %5.1f", run(k<BIG, 2, OP>, w, BIG ? 18 : 36)); printf(" %5.1f", run(k<BIG, 3, OP>, w, BIG ? 18 : 36)); \
    printf(" %5.1f", run(k<BIG, 4, OP>, w, BIG ? 18 : 36)); printf(" %5.1f", run(k<BIG, 6, OP>, w, BIG ? 18 : 36)); \
    printf(" %5.1f", run(k<BIG, 8, OP>, w, BIG ? 18 : 36)); } printf("\n"); }
int main() {
    printf("cycles (at 2.4 GHz) per MFMA with NV = 0 1 2 3 4 6 8 fillers after each, one / two waves per SIMD\n");
    ROW(0, 0, "16x16x32 + v_fmac_f32") ROW(0, 1, "16x16x32 + v_exp_f32") ROW(0, 2, "16x16x32 + v_cvt_pk_bf16_f32")
    ROW(0, 3, "16x16x32 + v_and_b32") ROW(0, 4, "16x16x32 + v_sub_f32") ROW(0, 5, "16x16x32 + v_rcp_f32")
    ROW(1, 0, "32x32x16 + v_fmac_f32") ROW(1, 1, "32x32x16 + v_exp_f32") ROW(1, 3, "32x32x16 + v_and_b32")
    return 0;
}
