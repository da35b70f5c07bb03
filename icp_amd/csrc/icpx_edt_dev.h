// Device-side LDT::Distance shared by the lookup kernel and the Go-ICP bound kernels.
#pragma once
#include <hip/hip_runtime.h>

// (int)(a / cell) for a >= 0 with the division rounded to nearest, as the reference computes it. The product with the
// rounded reciprocal is within 2e-7 (relative) of that quotient; unless it lies within 1e-6 of an integer both
// truncate to the same cell, and the division itself is only done for the few values that do.
static __device__ __forceinline__ int edt_cell_of(float a, float cell, float inv) {
    const float q = __fmul_rn(a, inv);
    if (fabsf(q - rintf(q)) > q * 1e-6f) return (int)q;
    return (int)__fdiv_rn(a, cell);
}

// LDT::Distance, src/testLDT.cpp:235-290: clamp (recording the overshoot in double), float index
// arithmetic (the overshoot too is a float difference widened to double), g * cellLen in float, + sqrt(overshoot^2) in double, result rounded to float.
static __device__ __forceinline__ float edt_distance(const int* __restrict__ sq, int div, float mnx, float mny, float mnz, float cell,
                                              float x, float y, float z) {
    double dx = 0.0, dy = 0.0, dz = 0.0;
    const float fdiv = (float)div;
    const float mxx = __fadd_rn(mnx, __fmul_rn(cell, fdiv));
    const float mxy = __fadd_rn(mny, __fmul_rn(cell, fdiv));
    const float mxz = __fadd_rn(mnz, __fmul_rn(cell, fdiv));
    if (x < mnx) { dx = (double)__fsub_rn(mnx, x); x = mnx; }
    if (x > mxx) { dx = (double)__fsub_rn(x, mxx); x = mxx; }
    if (y < mny) { dy = (double)__fsub_rn(mny, y); y = mny; }
    if (y > mxy) { dy = (double)__fsub_rn(y, mxy); y = mxy; }
    if (z < mnz) { dz = (double)__fsub_rn(mnz, z); z = mnz; }
    if (z > mxz) { dz = (double)__fsub_rn(z, mxz); z = mxz; }
    const float inv = __frcp_rn(cell);
    int ix = edt_cell_of(__fsub_rn(x, mnx), cell, inv);
    int iy = edt_cell_of(__fsub_rn(y, mny), cell, inv);
    int iz = edt_cell_of(__fsub_rn(z, mnz), cell, inv);
    ix = min(max(ix, 0), div - 1); iy = min(max(iy, 0), div - 1); iz = min(max(iz, 0), div - 1);
    const float g = __builtin_sqrtf((float)sq[((size_t)ix * div + iy) * div + iz]);
    const float inside = __fmul_rn(g, cell);
    // inside the box the overshoot is 0 and the double arithmetic below returns `inside` unchanged
    if (dx == 0.0 && dy == 0.0 && dz == 0.0) return inside;
    const double over = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    return (float)__dadd_rn((double)inside, __dsqrt_rn(over));
}
