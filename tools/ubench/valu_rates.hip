// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU forms used by the
// recurrent kernels.  Each kernel runs ITER iterations of 64 instructions on 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(NAME, BODY)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed) {           \
        float a[8], w = seed + threadIdx.x * 1e-3f, h = seed * 0.5f + threadIdx.x * 1e-4f;     \
        for (int i = 0; i < 8; ++i) a[i] = seed + i;                                           \
        for (int it = 0; it < iters; ++it) { REP64(BODY) }                                     \
        float s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + w + h;                                \
    }
#define B_FMA(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(h));
#define B_FMA_DPP(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(h), "v"(w));
#define B_ADD_DPP(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(h));
#define B_MOV_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(h));
#define B_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define B_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define B_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define B_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(w));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(w));
#define B_FMA3(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(h));
#define B_NOP(i) asm volatile("s_nop 1");
KERNEL(k_fmac, B_FMA) KERNEL(k_fmac_dpp, B_FMA_DPP) KERNEL(k_add_dpp, B_ADD_DPP) KERNEL(k_mov_dpp, B_MOV_DPP)
KERNEL(k_mul, B_MUL) KERNEL(k_exp, B_EXP) KERNEL(k_rcp, B_RCP) KERNEL(k_sqrt, B_SQRT) KERNEL(k_cndmask, B_CNDMASK)
KERNEL(k_mov, B_MOV) KERNEL(k_fma3, B_FMA3) KERNEL(k_nop, B_NOP)

// packed fp32 FMA: 2 FMAs per lane per instruction
__global__ __launch_bounds__(256) void k_pk_fma(float* out, int iters, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], w = {seed + threadIdx.x * 1e-3f, seed}, h = {seed * 0.5f, seed + threadIdx.x * 1e-4f};
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i, seed - i};
#define B_PK(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(h));
    for (int it = 0; it < iters; ++it) { REP64(B_PK) }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// dependent chain of DPP fmacs (latency)
__global__ __launch_bounds__(256) void k_fmac_dpp_dep(float* out, int iters, float seed) {
    float a = seed, w = seed + threadIdx.x * 1e-3f, h = seed * 0.5f;
#define B_DEP(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(h), "v"(w));
    for (int it = 0; it < iters; ++it) { REP64(B_DEP) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_fmac_dep(float* out, int iters, float seed) {
    float a = seed, w = seed + threadIdx.x * 1e-3f, h = seed * 0.5f;
#define B_DEP2(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(h), "v"(w));
    for (int it = 0; it < iters; ++it) { REP64(B_DEP2) }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// 3 interleaved dependent chains (the rotdot3 pattern)
__global__ __launch_bounds__(256) void k_fmac_dpp_dep3(float* out, int iters, float seed) {
    float a = seed, b = seed + 1, c = seed + 2, w = seed + threadIdx.x * 1e-3f, h = seed * 0.5f;
#define B_DEP3(i) asm volatile("v_fmac_f32_dpp %0, %3, %4 row_ror:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %3, %4 row_ror:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %2, %3, %4 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c) : "v"(h), "v"(w));
    for (int it = 0; it < iters; ++it) { REP8(B_DEP3) REP8(B_DEP3) B_DEP3(0) B_DEP3(0) B_DEP3(0) B_DEP3(0) B_DEP3(0) }  // 63 instrs
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;
}

template <typename K>
void run(const char* name, K k, int waves_per_simd, int ninstr = 64) {
    float* d; hipMalloc(&d, 1 << 26);
    int cus = 256, blocks = cus * waves_per_simd, iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // each SIMD hosts waves_per_simd waves, each issuing iters*ninstr instructions
    double instr_per_simd = (double)waves_per_simd * iters * ninstr;
    double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-16s waves/SIMD %d: %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz, %.2f @2.1GHz)\n", name, waves_per_simd,
           ns_per_instr, ns_per_instr * 2.4, ns_per_instr * 2.1);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run("v_fmac_f32", k_fmac, w); run("v_fma_f32", k_fma3, w); run("v_fmac_f32_dpp", k_fmac_dpp, w);
        run("v_add_f32_dpp", k_add_dpp, w); run("v_mov_b32_dpp", k_mov_dpp, w); run("v_mul_f32", k_mul, w);
        run("v_pk_fma_f32", k_pk_fma, w); run("v_exp_f32", k_exp, w); run("v_rcp_f32", k_rcp, w); run("v_sqrt_f32", k_sqrt, w);
        run("v_cndmask_b32", k_cndmask, w); run("v_mov_b32", k_mov, w); run("s_nop 1", k_nop, w);
        run("fmac_dpp dep1", k_fmac_dpp_dep, w); run("fmac dep1", k_fmac_dep, w); run("fmac_dpp dep3", k_fmac_dpp_dep3, w, 63);
        printf("\n");
    }
    return 0;
}
