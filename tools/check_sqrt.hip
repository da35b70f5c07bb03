// Exhaustive check of edt_sqrt_of_int (icp_amd/csrc/icpx_edt_dev.h) against the compiler's correctly rounded sqrtf for
// every integer 0 .. 2^23 (squared distances of the largest grid stay below 2^22).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I icp_amd/csrc -o build_ab/check_sqrt tools/check_sqrt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "icpx_edt_dev.h"
__global__ void k(unsigned n, unsigned* bad) {
    for (unsigned v = blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += gridDim.x * blockDim.x) {
        const float a = (float)v;
        if (__float_as_uint(edt_sqrt_of_int(a)) != __float_as_uint(__builtin_sqrtf(a))) atomicAdd(bad, 1u);
    }
}
int main() {
    unsigned* bad; unsigned h = 0;
    if (hipMalloc(&bad, 4) != hipSuccess) return 2;
    hipMemset(bad, 0, 4);
    k<<<1024, 256>>>(1u << 23, bad);
    if (hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("mismatches: %u\n", h);
    return h != 0;
}
