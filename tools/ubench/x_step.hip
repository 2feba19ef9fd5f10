// Micro-benchmark 6 (r05): the forward step of gru_s16x.hip in isolation — split of [h | features] into three bf16 terms, six
// pinned M tiles x six products of v_mfma_f32_16x16x32_bf16, gate arithmetic — at 1 .. 4 waves per SIMD, in variants:
//   V=0 as the kernel has it; V=1 no transcendental ops (rcp/exp -> fma); V=2 no split (operand constant); V=3 no MFMA (VALU only);
//   V=4 MFMAs only (no split, no gates)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pk(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);
    p3 = pk(sa, sb);
}
__device__ __forceinline__ f32x4 mf(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int V> __device__ __forceinline__ float sig(float v) {
    if (V == 1) return __builtin_fmaf(v, 0.25f, 0.5f);
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v) + 1.0f);
}
template <int V> __device__ __forceinline__ float th(float v) {
    if (V == 1) return __builtin_fmaf(v, 0.25f, 0.1f);
    return __builtin_fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v * 2.885f) + 1.0f), -2.0f, 1.0f);
}
template <int V, int THREADS, int SYNC = 0>
__global__ __launch_bounds__(THREADS) void k(const u32x4* __restrict__ tab, float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    u32x4 A[6][3];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[t][j] = tab[(t * 3 + j) * 64 + lane];
    float h[6], fs[2] = {seed * 0.3f, 1.0f};
    for (int j = 0; j < 6; ++j) h[j] = seed * 0.01f * (j + lane);
    u32x4 B[3] = {u32x4{1, 2, 3, 4}, u32x4{5, 6, 7, 8}, u32x4{9, 10, 11, 12}};
    for (int it = 0; it < iters; ++it) {
        if (V != 2 && V != 4) {
            float v[8] = {h[0], h[1], h[2], h[3], h[4], h[5], fs[0], fs[1]};
#pragma unroll
            for (int p = 0; p < 4; ++p) { unsigned a, b, c; split_pair(v[2 * p], v[2 * p + 1], a, b, c); B[0][p] = a; B[1][p] = b; B[2][p] = c; }
        }
        f32x4 acc[6];
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (SYNC == 1) __builtin_amdgcn_s_barrier();
        if (SYNC == 2 && (it & 1) == ((threadIdx.x >> 8) & 1)) __builtin_amdgcn_s_barrier();
        if (V != 3) {
#pragma unroll
            for (int t = 0; t < 6; t += 2) {
                f32x4 c = acc[t], d = acc[t + 1];
                c = mf(A[t][0], B[2], c); d = mf(A[t + 1][0], B[2], d);
                c = mf(A[t][2], B[0], c); d = mf(A[t + 1][2], B[0], d);
                c = mf(A[t][1], B[1], c); d = mf(A[t + 1][1], B[1], d);
                c = mf(A[t][0], B[1], c); d = mf(A[t + 1][0], B[1], d);
                c = mf(A[t][1], B[0], c); d = mf(A[t + 1][1], B[0], d);
                c = mf(A[t][0], B[0], c); d = mf(A[t + 1][0], B[0], d);
                acc[t] = c; acc[t + 1] = d;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[t] = f32x4{__uint_as_float(B[0][0]) * 1e-30f, __uint_as_float(B[1][1]) * 1e-30f, __uint_as_float(B[2][2]) * 1e-30f, __uint_as_float(B[0][3]) * 1e-30f};
        }
        if (V != 4) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int sr = j, sz = 6 + j, sh = 12 + j, si = 18 + j;
                const float r = sig<V>(acc[sr / 4][sr % 4]), z = sig<V>(acc[sz / 4][sz % 4]);
                const float nn = th<V>(__builtin_fmaf(r, acc[sh / 4][sh % 4], acc[si / 4][si % 4]));
                h[j] = __builtin_fmaf(z, h[j] - nn, nn);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t) B[t % 3][t % 4] ^= __float_as_uint(acc[t][t % 4]) & 1u;
        }
    }
    float s = 0;
    for (int j = 0; j < 6; ++j) s += h[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + B[0][0];
}
template <typename K>
double run(K kern, const u32x4* tab, int threads) {
    float* d; (void)hipMalloc(&d, 1 << 26);
    int iters = 2000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, tab, d, 50, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, tab, d, iters, 0.5f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(d);
    return ms * 1e-3 * 2.4e9 / ((double)(threads / 256) * iters);      // SIMD cycles per wave-step
}
#define ROW(V, NAME) printf("%-44s w1 %7.0f  w2 %7.0f  w3 %7.0f  w4 %7.0f\n", NAME, run(k<V, 256>, tab, 256), run(k<V, 512>, tab, 512), run(k<V, 768>, tab, 768), run(k<V, 1024>, tab, 1024));
int main() {
    u32x4* tab; (void)hipMalloc(&tab, 18 * 64 * 16);
    (void)hipMemset(tab, 0x3c, 18 * 64 * 16);
    printf("SIMD cycles (at 2.4 GHz) per wave-step (36 MFMA; ~150 VALU incl. 36 transcendental)\n");
    printf("%-44s w2 %7.0f  w4 %7.0f\n", "full step, barrier before the MFMAs", run(k<0, 512, 1>, tab, 512), run(k<0, 1024, 1>, tab, 1024));
    ROW(0, "full step") ROW(1, "no transcendentals") ROW(2, "no split") ROW(3, "no MFMA") ROW(4, "MFMA only")
    return 0;
}
