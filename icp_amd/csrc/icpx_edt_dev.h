// Device-side LDT::Distance shared by the lookup kernel and the Go-ICP bound kernels.
#pragma once
#include <hip/hip_runtime.h>

// LDT::Distance, src/testLDT.cpp:235-290: clamp (recording the overshoot), float index arithmetic, g * cellLen in
// float, + sqrt(overshoot^2) in double, result rounded to float.
//
// The cell of a coordinate is (int)(a / cell) with the division rounded to nearest, as the reference computes it. The
// product with the rounded reciprocal is within 2e-7 (relative) of that quotient; unless it lies within 1e-6 of an
// integer both truncate to the same cell. The division itself is done only in waves where some lane has such a value
// (a branch on the whole wave: left to the compiler, the rare case was turned into selects and every lane paid for
// three divisions on every call). The overshoot is kept as the float difference the reference forms and widened to
// double only outside the box; the voxel is read through a buffer descriptor with a 32-bit offset (a grid of at most
// 1023^3 voxels has fewer than 2^32 bytes).
// sqrtf of a voxel's squared distance (an integer below 2^22, as a float), correctly rounded like the reference's: the
// hardware's approximation (within one unit in the last place) and a step to the neighbour below or above where the
// remainder says so. This is the compiler's own expansion of a correctly rounded sqrtf without the rescaling of tiny
// arguments and the zero / infinity / NaN cases, none of which an integer in that range needs (0 comes out as 0: the
// neighbour below is NaN and compares false, the one above leaves a remainder that is not positive).
static __device__ __forceinline__ float edt_sqrt_of_int(float a) {
    float s = __builtin_amdgcn_sqrtf(a);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_below = __builtin_fmaf(-below, s, a), r_above = __builtin_fmaf(-above, s, a);
    if (r_below <= 0.f) s = below;
    if (r_above > 0.f) s = above;
    return s;
}

// One axis of LDT::Distance in two steps, for callers that look up several voxels per point. edt_axis_begin: the
// coordinate clamped into the box [mn, mx] (mx = mn + cell * div in float), the overshoot as the float difference the
// reference forms (mn - v below the box, v - mx above it, else 0; written as a maximum, which selects the same value),
// the offset a = v - mn and its quotient by the cell as a product with the rounded reciprocal `inv`. `near` says that
// the product may truncate to another cell than the correctly rounded quotient a / cell, which the caller then has to
// form (edt_axis_exact) before edt_axis_cell truncates and clamps it.
struct EdtAxis { float a, q, over; bool near; };
static __device__ __forceinline__ EdtAxis edt_axis_begin(float v, float mn, float mx, float inv) {
    EdtAxis r;
    r.over = fmaxf(fmaxf(__fsub_rn(mn, v), __fsub_rn(v, mx)), 0.f);
    v = fminf(fmaxf(v, mn), mx);
    r.a = __fsub_rn(v, mn);
    r.q = __fmul_rn(r.a, inv);
    r.near = !(fabsf(r.q - rintf(r.q)) > r.q * 1e-6f);
    return r;
}
// The same without the overshoot (`moved` says whether there is one, for the caller to form it where needed), and with a
// cheaper test for `near`: the fraction of the product lies within `eps` of 0 or 1, eps >= 1e-6 times the largest
// quotient (the number of cells per axis), so `near` is set whenever edt_axis_begin sets it.
static __device__ __forceinline__ EdtAxis edt_axis_begin_inside(float v, float mn, float mx, float inv, float eps, bool& moved) {
    EdtAxis r;
    r.over = 0.f;
    const float c = fminf(fmaxf(v, mn), mx);
    moved = c != v;                 // the coordinate lay outside [mn, mx]
    v = c;
    r.a = __fsub_rn(v, mn);
    r.q = __fmul_rn(r.a, inv);
    r.near = !(fabsf(__builtin_amdgcn_fractf(r.q) - 0.5f) < 0.5f - eps);
    return r;
}
// (a >= 0 and so is the quotient: only the upper end needs clamping)
static __device__ __forceinline__ unsigned edt_axis_cell_inside(const EdtAxis& r, int div) { return (unsigned)min((int)r.q, div - 1); }
static __device__ __forceinline__ void edt_axis_exact(EdtAxis& r, float cell) {
    if (r.near) r.q = __fdiv_rn(r.a, cell);
}
static __device__ __forceinline__ unsigned edt_axis_cell(const EdtAxis& r, int div) { return (unsigned)min(max((int)r.q, 0), div - 1); }

// The rest of LDT::Distance for a voxel given by its byte offset in the grid: g * cellLen in float, plus the length of
// the overshoot in double where there is one.
static __device__ __forceinline__ float edt_distance_at(__amdgpu_buffer_rsrc_t grid, unsigned byte_offset, float cell, float ox, float oy, float oz) {
    const float g = edt_sqrt_of_int((float)__builtin_amdgcn_raw_buffer_load_b32(grid, (int)byte_offset, 0, 0));
    const float inside = __fmul_rn(g, cell);
    // inside the box the overshoot is 0 and the double arithmetic below would return `inside` unchanged
    if (ox == 0.f && oy == 0.f && oz == 0.f) return inside;
    const double dx = (double)ox, dy = (double)oy, dz = (double)oz;
    const double over = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    return (float)__dadd_rn((double)inside, __dsqrt_rn(over));
}

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t edt_grid_rsrc(const int* sq) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(sq), 0, -1, 0x00020000);
}

static __device__ __forceinline__ float edt_distance(const int* __restrict__ sq, int div, int zs, float mnx, float mny, float mnz, float cell,
                                              float x, float y, float z) {
    const float fdiv = (float)div;
    const float mxx = __fadd_rn(mnx, __fmul_rn(cell, fdiv));
    const float mxy = __fadd_rn(mny, __fmul_rn(cell, fdiv));
    const float mxz = __fadd_rn(mnz, __fmul_rn(cell, fdiv));
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (x < mnx) { ox = __fsub_rn(mnx, x); x = mnx; }
    if (x > mxx) { ox = __fsub_rn(x, mxx); x = mxx; }
    if (y < mny) { oy = __fsub_rn(mny, y); y = mny; }
    if (y > mxy) { oy = __fsub_rn(y, mxy); y = mxy; }
    if (z < mnz) { oz = __fsub_rn(mnz, z); z = mnz; }
    if (z > mxz) { oz = __fsub_rn(z, mxz); z = mxz; }
    const float inv = __frcp_rn(cell);
    const float ax = __fsub_rn(x, mnx), ay = __fsub_rn(y, mny), az = __fsub_rn(z, mnz);
    float qx = __fmul_rn(ax, inv), qy = __fmul_rn(ay, inv), qz = __fmul_rn(az, inv);
    const bool cx = !(fabsf(qx - rintf(qx)) > qx * 1e-6f), cy = !(fabsf(qy - rintf(qy)) > qy * 1e-6f), cz = !(fabsf(qz - rintf(qz)) > qz * 1e-6f);
    if (__any(cx || cy || cz)) {
        if (cx) qx = __fdiv_rn(ax, cell);
        if (cy) qy = __fdiv_rn(ay, cell);
        if (cz) qz = __fdiv_rn(az, cell);
    }
    const int ix = min(max((int)qx, 0), div - 1), iy = min(max((int)qy, 0), div - 1), iz = min(max((int)qz, 0), div - 1);
    const unsigned voxel = ((unsigned)ix * (unsigned)div + (unsigned)iy) * (unsigned)zs + (unsigned)iz;       // rows of the grid are zs values apart
    const __amdgpu_buffer_rsrc_t grid = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(sq), 0, -1, 0x00020000);
    const float g = edt_sqrt_of_int((float)__builtin_amdgcn_raw_buffer_load_b32(grid, (int)(voxel * 4u), 0, 0));
    const float inside = __fmul_rn(g, cell);
    // inside the box the overshoot is 0 and the double arithmetic below would return `inside` unchanged
    if (ox == 0.f && oy == 0.f && oz == 0.f) return inside;
    const double dx = (double)ox, dy = (double)oy, dz = (double)oz;
    const double over = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
    return (float)__dadd_rn((double)inside, __dsqrt_rn(over));
}
