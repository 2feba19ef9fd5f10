// Probe of the register layout of v_mfma_f32_16x16x1_4b_f32 (the 4-block MFMA behind the gate-parallel train kernels' weight gradients):
// block k multiplies lanes 16k..16k+15 of A and B; register 4 k + r of lane l = A[16 k + 4 (l / 16) + r] * B[16 k + l % 16].
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_16x16x1_4b_layout.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* o, const float* a, const float* b) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) o[threadIdx.x * 16 + i] = acc[i];
}
int main() {
    float ha[64], hb[64], ho[1024];
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + l; }
    float *a, *b, *o;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 4096);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(o, a, b);
    hipMemcpy(ho, o, 4096, hipMemcpyDeviceToHost);
    // hypothesis: reg 4*blk + r of lane l = a[16*blk + 4*(l/16) + r] * b[16*blk + l%16]
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int blk = 0; blk < 4; ++blk) for (int r = 0; r < 4; ++r) {
        float want = ha[16 * blk + 4 * (l / 16) + r] * hb[16 * blk + l % 16];
        if (ho[l * 16 + 4 * blk + r] != want) ++bad;
    }
    printf("hypothesis A mismatches: %d\n", bad);
    for (int l = 0; l < 64; l += 17) { printf("lane %d:", l); for (int i = 0; i < 16; ++i) printf(" %.0f", ho[l * 16 + i]); printf("\n"); }
    return 0;
}
