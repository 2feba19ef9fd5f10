// Micro-benchmark (r06): does the OPERAND FORM of a vector instruction change its issue cost?  All-VGPR v_fmac_f32 / v_mul_f32 issue in ~1.1 - 1.2 ns
// per wave64 instruction per SIMD on an MI355X, the same instructions with a scalar (SGPR) source, VOP3-only instructions and DPP forms in ~1.75.
// Each kernel runs ITER iterations of 64 instructions on 8 independent chains, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define KERNEL(NAME, BODY)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed, float sc) { \
        float a[8], w = seed + threadIdx.x * 1e-3f, h = seed * 0.5f + threadIdx.x * 1e-4f;     \
        int iw = threadIdx.x;                                                                  \
        for (int i = 0; i < 8; ++i) a[i] = seed + i;                                           \
        for (int it = 0; it < iters; ++it) { REP64(BODY) }                                     \
        float s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + w + h + (float)iw;                    \
    }
#define B_MUL_V(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define B_MUL_S(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
#define B_MUL_K(i) asm volatile("v_mul_f32 %0, 0x3f7fbe77, %0" : "+v"(a[i]));
#define B_MUL_I(i) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(a[i]));
#define B_ADD_V(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define B_ADD_S(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
#define B_FMAC_S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(w));
#define B_FMA_S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(w));
#define B_FMA_V(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(h), "v"(w));
#define B_MED3_V(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(h), "v"(w));
#define B_MED3_S(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(w));
#define B_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
#define B_CVT_FI(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
#define B_CVT_IF(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define B_SUB_V(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define B_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
#define B_AND(i) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a[i]));
#define B_ANDV(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(iw));
#define B_CVTPK(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define B_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(iw));
#define B_CNDE64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "s"(msk));
#define B_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(iw));
#define B_MAX_V(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(w));
#define B_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
KERNEL(k_mul_v, B_MUL_V) KERNEL(k_mul_s, B_MUL_S) KERNEL(k_mul_k, B_MUL_K) KERNEL(k_mul_i, B_MUL_I) KERNEL(k_add_v, B_ADD_V) KERNEL(k_add_s, B_ADD_S)
KERNEL(k_fmac_s, B_FMAC_S) KERNEL(k_fma_s, B_FMA_S) KERNEL(k_fma_v, B_FMA_V) KERNEL(k_med3_v, B_MED3_V) KERNEL(k_med3_s, B_MED3_S)
KERNEL(k_rndne, B_RNDNE) KERNEL(k_cvt_fi, B_CVT_FI) KERNEL(k_cvt_if, B_CVT_IF) KERNEL(k_sub_v, B_SUB_V) KERNEL(k_lshl, B_LSHL) KERNEL(k_and, B_AND)
KERNEL(k_andv, B_ANDV) KERNEL(k_cvtpk, B_CVTPK) KERNEL(k_lshladd, B_LSHLADD) KERNEL(k_perm, B_PERM) KERNEL(k_max_v, B_MAX_V) KERNEL(k_exp, B_EXP)

template <typename K>
void run(const char* name, K k, int w = 4) {
    float* d; hipMalloc(&d, 1 << 26);
    const int blocks = 256 * w, iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f, 0.9999f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.9999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s waves/SIMD %d: %.3f ns per wave-instr per SIMD\n", name, w, ms * 1e6 / ((double)w * iters * 64));
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
    run("v_mul_f32 v, v, v", k_mul_v, w); run("v_mul_f32 v, s, v", k_mul_s, w); run("v_mul_f32 v, literal, v", k_mul_k, w); run("v_mul_f32 v, 2.0, v", k_mul_i, w);
    run("v_add_f32 v, v, v", k_add_v, w); run("v_add_f32 v, s, v", k_add_s, w); run("v_sub_f32 v, v, v", k_sub_v, w);
    run("v_fmac_f32 v, s, v", k_fmac_s, w); run("v_fma_f32 v, v, s, v", k_fma_s, w); run("v_fma_f32 v, v, v, v", k_fma_v, w);
    run("v_med3_f32 v, v, v, v", k_med3_v, w); run("v_med3_f32 v, v, s, v", k_med3_s, w); run("v_max_f32 v, v, v", k_max_v, w);
    run("v_rndne_f32", k_rndne, w); run("v_cvt_f32_i32", k_cvt_fi, w); run("v_cvt_i32_f32", k_cvt_if, w);
    run("v_lshlrev_b32 v, 1, v", k_lshl, w); run("v_and_b32 v, literal, v", k_and, w); run("v_and_b32 v, v, v", k_andv, w);
    run("v_cvt_pk_bf16_f32", k_cvtpk, w); run("v_lshl_add_u32", k_lshladd, w); run("v_perm_b32", k_perm, w); run("v_exp_f32", k_exp, w);
    printf("\n");
    }
    return 0;
}
