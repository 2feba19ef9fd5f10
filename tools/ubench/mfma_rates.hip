// Micro-benchmark 2: f32 MFMA issue cost alone and next to VALU work, permlane swaps, cndmask forms,
// LDS read/write forms — inputs for the 16-sequences-per-wave MFMA design.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float seed) {
    f32x4 c[8]; float a = seed + threadIdx.x, b = seed * 0.5f;
    for (int i = 0; i < 8; ++i) c[i] = f32x4{seed, 0, 0, 0};
#define B_M(i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    for (int it = 0; it < iters; ++it) { REP64(B_M) }
    float s = 0; for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 1 MFMA + NV plain VALU fmacs per group: do they overlap?
template <int NV>
__global__ __launch_bounds__(256) void k_mfma_valu(float* out, int iters, float seed) {
    f32x4 c[4]; float a = seed + threadIdx.x, b = seed * 0.5f, v[8];
    for (int i = 0; i < 4; ++i) c[i] = f32x4{seed, 0, 0, 0};
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            c[g & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[g & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j & 7]) : "v"(a), "v"(b));
        }
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += c[i][0]; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define KERNEL(NAME, BODY)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed) {           \
        float a[8], w = seed + threadIdx.x * 1e-3f, h = seed * 0.5f + threadIdx.x * 1e-4f;     \
        for (int i = 0; i < 8; ++i) a[i] = seed + i;                                           \
        for (int it = 0; it < iters; ++it) { REP64(BODY) }                                     \
        float s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + w + h;                                \
    }
#define B_CND_VCC(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(w), "v"(h));
#define B_CND_E64(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a[i]) : "v"(w), "v"(h) : "s20", "s21");
#define B_CND_DEP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(w) : "s20", "s21");
#define B_PL16(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
#define B_PL32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 7]));
#define B_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define B_CMP(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(w) : "vcc");
KERNEL(k_cnd_vcc, B_CND_VCC) KERNEL(k_cnd_e64, B_CND_E64) KERNEL(k_cnd_dep, B_CND_DEP) KERNEL(k_pl16, B_PL16) KERNEL(k_pl32, B_PL32)
KERNEL(k_max, B_MAX) KERNEL(k_cmp, B_CMP)

// LDS forms
__global__ __launch_bounds__(256) void k_lds(float* out, int iters, float seed, int mode) {
    __shared__ __attribute__((aligned(16))) float sm[256 * 8];
    for (int i = threadIdx.x; i < 256 * 8; i += 256) sm[i] = seed + i;
    __syncthreads();
    float acc = 0; const int t = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (mode == 0) { float4 v = *reinterpret_cast<float4*>(&sm[((t + u * 16) & 255) * 4]); acc += v.x + v.w; }
            else if (mode == 1) { acc += sm[(t + u * 17) & 2047]; }
            else if (mode == 2) { sm[(t + u * 17) & 2047] = acc; acc += 1.0f; }
            else { float2 v = *reinterpret_cast<float2*>(&sm[((t >> 4) * 66 + u * 2) & 2046]); acc += v.x + v.y; }
        }
    }
    out[blockIdx.x * blockDim.x + t] = acc;
}

template <typename K, typename... A>
void run(const char* name, K k, int wps, int ninstr, A... extra) {
    float* d; hipMalloc(&d, 1 << 26);
    int blocks = 256 * wps, iters = 10000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, extra...);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, extra...);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ns = ms * 1e6 / ((double)wps * iters * ninstr);
    printf("%-28s waves/SIMD %d: %.3f ns per wave-instr per SIMD (%.1f cycles @2.2GHz)\n", name, wps, ns, ns * 2.2);
    hipFree(d);
}
// permlane semantic probe
__global__ void k_probe(int* out) {
    int a = threadIdx.x, b = 100 + threadIdx.x;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b;
    int c = threadIdx.x, d = 100 + threadIdx.x;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    out[128 + threadIdx.x] = c; out[192 + threadIdx.x] = d;
}
int main() {
    for (int w : {1, 2, 4}) {
        run("mfma_f32_16x16x4", k_mfma, w, 64);
        run("mfma + 4 fmac (per group)", k_mfma_valu<4>, w, 16);
        run("mfma + 8 fmac (per group)", k_mfma_valu<8>, w, 16);
        run("mfma + 12 fmac (per group)", k_mfma_valu<12>, w, 16);
        run("mfma + 16 fmac (per group)", k_mfma_valu<16>, w, 16);
        run("v_cndmask vcc (indep)", k_cnd_vcc, w, 64); run("v_cndmask e64 sgpr (indep)", k_cnd_e64, w, 64);
        run("v_cndmask e64 (dep dst=src0)", k_cnd_dep, w, 64);
        run("v_permlane16_swap", k_pl16, w, 64); run("v_permlane32_swap", k_pl32, w, 64);
        run("v_max_f32", k_max, w, 64); run("v_cmp_gt_f32", k_cmp, w, 64);
        run("ds_read_b128", k_lds, w, 16, 0); run("ds_read_b32", k_lds, w, 16, 1); run("ds_write_b32", k_lds, w, 16, 2);
        run("ds_read_b64 bcast", k_lds, w, 16, 3);
        printf("\n");
    }
    int* d; hipMalloc(&d, 256 * 4); hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"pl16 a", "pl16 b", "pl32 c", "pl32 d"};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 8) printf(" %d", h[r * 64 + i]); printf("\n"); }
    return 0;
}
