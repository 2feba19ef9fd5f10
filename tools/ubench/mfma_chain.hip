// Micro-benchmark 4: exact-fp32 MFMA issue interval vs the number of independent in-place accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(256) void k_chain(float* out, int iters, float seed) {
    f32x4 c[D];
    float a = seed + threadIdx.x * 1e-3f, b = seed * 0.5f;
    for (int i = 0; i < D; ++i) c[i] = f32x4{seed, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 48; ++g) c[g % D] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[g % D], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < D; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K k, int wps) {
    float* d; hipMalloc(&d, 1 << 26);
    const int blocks = 256 * wps, iters = 5000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s waves/SIMD %d : %6.2f ns per MFMA per SIMD\n", name, wps, ms * 1e6 / iters / 48 / wps);
    (void)hipFree(d);
}
int main() {
    for (int wps = 1; wps <= 2; ++wps) {
        run("1 chain", k_chain<1>, wps); run("2 chains", k_chain<2>, wps); run("3 chains", k_chain<3>, wps);
        run("4 chains", k_chain<4>, wps); run("6 chains", k_chain<6>, wps); run("8 chains", k_chain<8>, wps);
    }
    return 0;
}
