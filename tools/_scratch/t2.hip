#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../opendpd_amd/csrc/odpd_device.h"
using namespace odpd;
__global__ void k(float* o) {
    const float t = 100.0f + threadIdx.x, f = -1.0f;
    for (int r = 0; r < 4; ++r) o[r * 64 + threadIdx.x] = vsel(kRowMask[r], t, f);
    o[256 + threadIdx.x] = vsel(kRowMask[1], 7.0f, vsel(kRowMask[2], 0.0f, t));
}
int main() {
    float* o; hipMalloc(&o, 320 * 4);
    k<<<1, 64>>>(o);
    float h[320]; hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    for (int r = 0; r < 5; ++r) { printf("case %d:", r); for (int l = 0; l < 64; l += 8) printf(" %.0f", h[r * 64 + l]); printf("\n"); }
    return 0;
}
