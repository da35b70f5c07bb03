// Exhaustive check of the reciprocal-based floor division the envelope scan uses (icp_amd/csrc/icpx_edt.hip,
// floordiv_pos and the instruction sequence of the hand-written scan): for every numerator 0 <= N < 2^22 and every even divisor 2 <= D <= 2046 the result equals
// N / D whenever that quotient is below 1400, and is at least 1023 otherwise. Build and run on the GPU:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/check_floordiv tools/check_floordiv.hip && /tmp/check_floordiv
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ int floordiv_pos(int N, int D) {
    return (int)((float)(2 * N + 1) * __builtin_amdgcn_rcpf((float)(2 * D)));
}

// The same quotient as the hand-written scan computes it: dd = -N, d = D / 2, result w = 1 + floor(N / D).
__device__ __forceinline__ int crossing_asm(int dd, int d) {
    int x, y, w;
    asm volatile(
        "v_mad_i32_i24 %[x], %[dd], -2, 1\n"
        "v_lshlrev_b32 %[y], 2, %[d]\n"
        "v_cvt_f32_i32 %[x], %[x]\n"
        "v_cvt_f32_i32 %[y], %[y]\n"
        "v_rcp_f32 %[y], %[y]\n"
        "s_nop 0\n"
        "v_mul_f32 %[x], %[x], %[y]\n"
        "v_cvt_i32_f32 %[x], %[x]\n"
        "v_add_u32 %[w], 1, %[x]\n"
        : [x] "=&v"(x), [y] "=&v"(y), [w] "=&v"(w) : [dd] "v"(dd), [d] "v"(d));
    return w;
}

__global__ void k_check(unsigned long long* bad) {
    const int D = 2 * (blockIdx.y + 1);
    unsigned long long wrong = 0;
    for (int N = blockIdx.x * blockDim.x + threadIdx.x; N < (1 << 22); N += gridDim.x * blockDim.x) {
        const int want = N / D, got = floordiv_pos(N, D), got2 = crossing_asm(-N, D / 2) - 1;
        wrong += want < 1400 ? got != want : got < 1023;
        wrong += want < 1400 ? got2 != want : got2 < 1023;
    }
    if (wrong) atomicAdd(bad, wrong);
}

int main() {
    unsigned long long* bad; unsigned long long h = 0;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMemset(bad, 0, 8) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_check, dim3(256, 1023), dim3(256), 0, 0, bad);
    if (hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("floordiv_pos: %llu mismatches over 2^22 numerators x 1023 divisors\n", h);
    return h != 0;
}
