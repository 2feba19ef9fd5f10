// Micro-benchmark 5 (r05): does a ds_read_b128 stream slow a bf16 MFMA (+ VALU) stream on the same SIMD?
// per group: 12 MFMAs on 2 accumulators (+ NV VALU after each) and NL ds_read_b128 (results unused or used as the A operand of the NEXT group).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BAR __builtin_amdgcn_sched_barrier(0);
// USE: 0 = loads are dead (still must complete), 1 = loaded values are the A operands of the next group (distinct registers per MFMA)
template <int NL, int NV, int USE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    extern __shared__ u32x4 sm[];
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) sm[i] = u32x4{(unsigned)i, 2u, 3u, 4u};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 c[2] = {f32x4{seed, 0, 0, 0}, f32x4{seed, 0, 0, 0}};
    u32x4 a[6], b = {5, 6, threadIdx.x, 8}, nx[6];
    for (int i = 0; i < 6; ++i) a[i] = u32x4{threadIdx.x + i, 2, 3, 4};
    float v[8], x = seed + threadIdx.x, y = seed * 0.5f;
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    unsigned keep = 0;
    for (int it = 0; it < iters; ++it) {
        const u32x4* p = sm + ((it & 7) * 6) * 64 + lane;
#pragma unroll
        for (int l = 0; l < NL; ++l) nx[l] = p[l * 64];
        BAR
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            c[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m % 6]), __builtin_bit_cast(bf16x8, b), c[m & 1], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j & 7]) : "v"(x), "v"(y));
        }
        BAR
#pragma unroll
        for (int l = 0; l < NL; ++l) { if (USE) a[l] = nx[l]; else keep ^= nx[l][0]; }
    }
    float s = c[0][0] + c[1][3] + keep;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
double run(K kern, int wps) {
    float* d; (void)hipMalloc(&d, 1 << 26);
    int iters = 4000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 65536, 0, d, 100, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 65536, 0, d, iters, 1.0f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(d);
    return ms * 1e-3 * 2.4e9 / ((double)wps * iters * 12);
}
#define ROW(NV, USE) { printf("NV=%d USE=%d:", NV, USE); for (int w : {1, 2}) { printf(" | w%d:", w); \
    printf(" %5.1f", run(k<0, NV, USE>, w)); printf(" %5.1f", run(k<1, NV, USE>, w)); printf(" %5.1f", run(k<2, NV, USE>, w)); \
    printf(" %5.1f", run(k<3, NV, USE>, w)); printf(" %5.1f", run(k<6, NV, USE>, w)); } printf("\n"); }
int main() {
    printf("cycles (at 2.4 GHz) per MFMA; per 12 MFMAs NL = 0 1 2 3 6 ds_read_b128; one / two waves per SIMD\n");
    ROW(0, 0) ROW(0, 1) ROW(2, 0) ROW(2, 1) ROW(4, 1)
    return 0;
}
