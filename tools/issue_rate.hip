// Issue-rate microbenchmark for one SIMD of gfx950: cycles per instruction for VALU-only, SALU-only, interleaved and
// dependent-VALU streams at 1..8 waves per SIMD. Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_rate tools/issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define V8 "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
#define VD8 "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
#define S8 "s_add_u32 s20, s20, s21\n s_add_u32 s21, s21, s22\n s_add_u32 s22, s22, s23\n s_add_u32 s23, s23, s20\n s_add_u32 s20, s20, s21\n s_add_u32 s21, s21, s22\n s_add_u32 s22, s22, s23\n s_add_u32 s23, s23, s20\n"
#define VS8 "v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, s21\n v_add_u32 %1, %1, %2\n s_add_u32 s21, s21, s22\n v_add_u32 %2, %2, %3\n s_add_u32 s22, s22, s23\n v_add_u32 %3, %3, %0\n s_add_u32 s23, s23, s20\n v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, s21\n v_add_u32 %1, %1, %2\n s_add_u32 s21, s21, s22\n v_add_u32 %2, %2, %3\n s_add_u32 s22, s22, s23\n v_add_u32 %3, %3, %0\n s_add_u32 s23, s23, s20\n"
template <int MODE>
__global__ void k(int* out, int iters) {
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(V8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (MODE == 1) asm volatile(S8 ::: "s20", "s21", "s22", "s23", "scc");
        if (MODE == 2) asm volatile(VS8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s20", "s21", "s22", "s23", "scc");
        if (MODE == 3) asm volatile(VD8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    if (a0 + a1 + a2 + a3 == 0x12345) out[0] = 1;
}
template <int MODE> float run(int* d, int wgs, int threads, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    int* d; (void)hipMalloc(&d, 4);
    const int iters = 20000;
    for (int wps : {1, 2, 3, 4, 8}) {   // waves per SIMD: workgroup of 256 threads = 4 waves = 1 per SIMD
        const int wgs = 256 * wps;
        float v = run<0>(d, wgs, 256, iters), s = run<1>(d, wgs, 256, iters), b = run<2>(d, wgs, 256, iters), vd = run<3>(d, wgs, 256, iters);
        const double c = 1e-3 * 2.4e9 / (8.0 * iters * wps);
        printf("waves/SIMD %d: cycles(2.4 GHz) per 8-group element per SIMD: valu %.2f  salu %.2f  valu+salu pair %.2f  dependent valu %.2f\n", wps, v * c, s * c, b * c, vd * c);
    }
    return 0;
}
