// Probe of the operand layout of v_mfma_i32_16x16x32_i8 (the integer matrix pipe behind the W8A8 mat-vecs of csrc/qat_s16.hip) and of its
// issue cost next to v_mfma_f32_16x16x4_f32.  Hypothesis: lane l feeds A[l % 16][8 (l / 16) + j] and B[8 (l / 16) + j][l % 16] as byte j of
// its 64-bit operand (j = 0..7), and holds D[4 (l / 16) + i][l % 16], i = 0..3 — the same D layout as the 16x16x4 f32 instruction.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_i32_16x16x32_i8_layout.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* o, const long* a, const long* b) {
    i32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) o[threadIdx.x * 4 + i] = acc[i];
}
__global__ void timing(long long* cyc, int n) {
    i32x4 ai = {0, 0, 0, 0};
    f32x4 af = {0, 0, 0, 0};
    long x = threadIdx.x * 0x0101010101010101L;
    float y = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) ai = __builtin_amdgcn_mfma_i32_16x16x32_i8(x, x, ai, 0, 0, 0);
    }
    long long t1 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) af = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, af, 0, 0, 0);
    }
    long long t2 = clock64();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = ai[0] + (int)af[0]; }
}
int main() {
    int8_t A[16][32], B[32][16];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) { A[i][kk] = (int8_t)((i * 7 + kk * 3) % 23 - 11); B[kk][i] = (int8_t)((i * 5 + kk * 11) % 19 - 9); }
    long ha[64], hb[64];
    for (int l = 0; l < 64; ++l) {
        uint64_t va = 0, vb = 0;
        for (int j = 0; j < 8; ++j) { va |= (uint64_t)(uint8_t)A[l % 16][8 * (l / 16) + j] << (8 * j); vb |= (uint64_t)(uint8_t)B[8 * (l / 16) + j][l % 16] << (8 * j); }
        ha[l] = (long)va; hb[l] = (long)vb;
    }
    long *a, *b; int* o; int ho[256];
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&o, 1024);
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(o, a, b);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        int row = 4 * (l / 16) + i, col = l % 16, want = 0;
        for (int kk = 0; kk < 32; ++kk) want += (int)A[row][kk] * (int)B[kk][col];
        if (ho[l * 4 + i] != want) ++bad;
    }
    printf("layout hypothesis mismatches: %d of 256\n", bad);
    long long* cyc; long long hc[3];
    hipMalloc(&cyc, 24);
    timing<<<1, 64>>>(cyc, 1000);
    hipMemcpy(hc, cyc, 24, hipMemcpyDeviceToHost);
    printf("dependent chain, cycles per instruction (s_memtime clock): i32_16x16x32_i8 %.2f   f32_16x16x4_f32 %.2f\n", hc[0] / 16000.0, hc[1] / 16000.0);
    return 0;
}
