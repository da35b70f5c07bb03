// Go-ICP bound evaluation: batched upper / lower bounds of translation cubes under given rotations.
//
// Stands in for the point loops of GoIcp::innerBnB (src/GoIcp.cpp:384-431), the rotation of the data
// cloud in OuterBnB (:264-272), GoIcp::Init's uncertainty table (:459-480) and the SSE loop of runICP
// (:113-135). One launch evaluates a whole batch of (rotation, translation cube, level) items; each
// item is split over several workgroups so that small batches still fill the chip, and a second kernel
// adds the per-workgroup partial sums in a fixed order.
//
// Arithmetic follows the reference per point in float: x' = (R11 x + R12 y) + R13 z, centre = corner +
// w/2, d = Distance(x' + centre) - maxRotDist, clamped at 0 (built with -ffp-contract=off). The sums
// are accumulated in double on the device; the per-point d can be written out for callers that need
// the reference's sequential float sums.
#include "icpx_ctx.h"
#include "icpx_edt_dev.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

#define GOICP_PI 3.1415926536     // src/GoIcp.cpp:17
#define GOICP_SQRT3 1.732050808   // src/GoIcp.cpp:18

struct EdtView {
    const int* __restrict__ sq;
    int div;
    int zs;                                      // values between consecutive rows of sq (z runs fastest)
    float mnx, mny, mnz, cell;
    const unsigned short* __restrict__ bricks;   // the same values in bricks of 4 x 4 x 4 voxels (see k_edt_bricks), or NULL
    int nb;                                      // bricks per axis
    const float* __restrict__ fbricks;           // the distances themselves (float) in the same brick order (see k_edt_fbricks), or NULL
};

// norm in float (the SQ() macro multiplies floats and std::sqrt picks the float overload,
// src/GoIcp.cpp:464-466), then (float)(c_level * (double)norm) with c_level = 2 sin(tmp/2) from the host.
struct RotConsts { double c[ICPX_MAXROTLEVEL]; };

__global__ void k_maxrotdist(const float4* __restrict__ data, size_t n, RotConsts rc, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = data[i];
    const float nrm = __builtin_sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(p.x, p.x), __fmul_rn(p.y, p.y)), __fmul_rn(p.z, p.z)));
#pragma unroll
    for (int l = 0; l < ICPX_MAXROTLEVEL; l++) out[(size_t)l * n + i] = (float)__dmul_rn(rc.c[l], (double)nrm);
}

// Bounds of one item over one slice of the cloud: every thread adds its points lo + t, lo + t + 256, ... in that order,
// then the wave's 64 sums by a butterfly, then the four waves as (0 + 1) + (2 + 3). k_bnb_eval8 adds in the same order.
static __device__ __forceinline__ void bnb_eval_item(const EdtView& e, const float4* __restrict__ data, size_t n, const float* __restrict__ maxrot,
                                                     const icpx_bnb_item* __restrict__ items, int item, int split, int nsplit,
                                                     double* __restrict__ partials, float* __restrict__ mindis, double (*sm)[4]) {
    const icpx_bnb_item it = items[item];
    const float half = it.w / 2;
    const float tx = it.x + half, ty = it.y + half, tz = it.z + half;
    const float maxTransDis = (float)(GOICP_SQRT3 / 2.0 * (double)it.w);
    // items that arrive in device memory cannot be checked by the host: a level beyond the table reads nothing and
    // yields NaN bounds
    const bool bad_level = it.level >= ICPX_MAXROTLEVEL;
    const float* mr = it.level >= 0 && !bad_level ? maxrot + (size_t)it.level * n : nullptr;
    const size_t per = (n + nsplit - 1) / nsplit;
    const size_t lo = (size_t)split * per, hi = lo + per < n ? lo + per : n;
    double ub = 0.0, lb = 0.0;
    // the next point and its rotation uncertainty are read while this one is looked up
    size_t i = lo + threadIdx.x;
    float4 p_next = i < hi ? data[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float mr_next = mr && i < hi ? mr[i] : 0.f;
    for (; i < hi; i += blockDim.x) {
        const float4 p = p_next;
        const float mr_i = mr_next;
        if (i + blockDim.x < hi) { p_next = data[i + blockDim.x]; if (mr) mr_next = mr[i + blockDim.x]; }
        const float rx = it.R[0] * p.x + it.R[1] * p.y + it.R[2] * p.z;
        const float ry = it.R[3] * p.x + it.R[4] * p.y + it.R[5] * p.z;
        const float rz = it.R[6] * p.x + it.R[7] * p.y + it.R[8] * p.z;
        float d = edt_distance(e.sq, e.div, e.zs, e.mnx, e.mny, e.mnz, e.cell, rx + tx, ry + ty, rz + tz);
        if (mr) d -= mr_i;
        if (d < 0) d = 0;
        if (mindis) mindis[(size_t)item * n + __float_as_int(p.w)] = d;
        ub += (double)(d * d);
        const float dis = d - maxTransDis;
        if (dis > 0) lb += (double)(dis * dis);
    }
    for (int o = 32; o > 0; o >>= 1) { ub += __shfl_xor(ub, o); lb += __shfl_xor(lb, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                      // a previous item's sums have been read
    if (lane == 0) { sm[0][wave] = ub; sm[1][wave] = lb; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const double v = (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
        partials[((size_t)item * nsplit + split) * 2 + threadIdx.x] = bad_level ? (double)NAN : v;
    }
}

__global__ __launch_bounds__(256) void k_bnb_eval(EdtView e, const float4* __restrict__ data, size_t n, const float* __restrict__ maxrot,
                                                  const icpx_bnb_item* __restrict__ items, int nsplit, double* __restrict__ partials,
                                                  float* __restrict__ mindis) {
    __shared__ double sm[2][4];
    bnb_eval_item(e, data, n, maxrot, items, blockIdx.x, blockIdx.y, nsplit, partials, mindis, sm);
}

// The squared distances once more as 16-bit values in bricks of 4 x 4 x 4 voxels, one 128-byte line each: element
// ((bx nb + by) nb + bz) 64 + (x & 3) 16 + (y & 3) 4 + (z & 3) with b = voxel >> 2. The 64 neighbouring points of a wave
// fall into a few bricks where they touch some sixty lines of the [x][y][z] grid, and the bound kernel is limited by
// the number of lines its lookups touch. Values of 65535 and more are stored as 65535, which sends the reader to the
// 32-bit grid (voxels more than 255 cells from the surface).
#define EDT_BRICK_ESCAPE 65535u
__global__ void k_edt_bricks(const int* __restrict__ sq, int div, int zs, int nb, unsigned short* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)nb * nb * nb * 64;
    if (e >= total) return;
    const unsigned b = (unsigned)(e >> 6), in = (unsigned)(e & 63);
    const unsigned bz = b % nb, by = b / nb % nb, bx = b / nb / nb;
    const unsigned x = bx * 4 + (in >> 4), y = by * 4 + (in >> 2 & 3), z = bz * 4 + (in & 3);
    unsigned v = EDT_BRICK_ESCAPE;
    if (x < (unsigned)div && y < (unsigned)div && z < (unsigned)div) {
        const unsigned w = (unsigned)sq[((size_t)x * div + y) * zs + z];
        v = w < EDT_BRICK_ESCAPE ? w : EDT_BRICK_ESCAPE;
    }
    out[e] = (unsigned short)v;
}

// The bricks once more with the distance itself in place of the squared cell count: sqrt(v) * cell, rounded as
// LDT::Distance rounds it (edt_distance_at), so that the bound kernel reads what it would otherwise compute for every
// lookup (a conversion, a square root made exact, a product: a third of its instructions). 256 bytes per brick.
__global__ void k_edt_fbricks(const int* __restrict__ sq, int div, int zs, int nb, float cell, float* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)nb * nb * nb * 64;
    if (e >= total) return;
    const unsigned b = (unsigned)(e >> 6), in = (unsigned)(e & 63);
    const unsigned bz = b % nb, by = b / nb % nb, bx = b / nb / nb;
    const unsigned x = bx * 4 + (in >> 4), y = by * 4 + (in >> 2 & 3), z = bz * 4 + (in & 3);
    float v = 0.f;
    if (x < (unsigned)div && y < (unsigned)div && z < (unsigned)div)
        v = __fmul_rn(edt_sqrt_of_int((float)sq[((size_t)x * div + y) * zs + z]), cell);
    out[e] = v;
}

// One step of the butterfly over 2 * CNT values per lane: the lanes with bit `o` set keep the upper CNT values and
// hand the lower ones to their partner l ^ o, the others the other way round; each adds what it receives.
template <int CNT>
static __device__ __forceinline__ void bnb_fold(double* v, int lane, int o) {
    const bool upper = (lane & o) != 0;
#pragma unroll
    for (int k = 0; k < CNT; k++) {
        const double send = upper ? v[k] : v[k + CNT];
        const double keep = upper ? v[k + CNT] : v[k];
        v[k] = keep + __shfl_xor(send, o);
    }
}

static __device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }

// Eight items per workgroup. The eight children of a translation cube (src/GoIcp.cpp:378-390) share the rotation, the
// level and the width, and each of their coordinates takes one of two values: the rotated point, its rotation
// uncertainty and the six (coordinate, axis) cell computations are then done once for all eight, and only the voxel
// lookup and the bound arithmetic per child. Which items form such a family is read off the items themselves (a
// comparison of bits, the same in every lane); any other eight items, and the last items of a batch whose size is no
// multiple of eight, are evaluated one after the other as k_bnb_eval does. Every value is computed by the same float
// operations on the same operands as there and added in the same order, so the bounds are the same bits.
// PARENTS: `items` holds one record per family, the parent cube, and the eight children are formed here as the search
// forms them (src/GoIcp.cpp:378-390: w' = w / 2, corner + (0 or 1) w' per axis, in float).
// A partial sum on its way out. AGENT: for a reader in another workgroup of the same launch, which may run on another
// XCD with an L2 of its own: an atomic exchange at agent scope is performed where all XCDs see it, and its returned
// value says that it has been (a plain store followed by a fence costs a write-back of the whole L2 per workgroup:
// six times the run time of the search).
template <bool AGENT>
static __device__ __forceinline__ void bnb_put(double* where, double v) {
    if (AGENT) {
        const double old = __hip_atomic_exchange(where, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" :: "v"(old));           // the value is waited for
    } else *where = v;
}

template <int BRICKS, bool PARENTS, int AHEAD = 1, bool AGENT_OUT = false>      // BRICKS: 0 the [x][y][z] grid, 1 the 16-bit bricks, 2 the float bricks
static __device__ __forceinline__ void bnb_eval8_body(const EdtView& e, const float4* __restrict__ data, size_t n, const float* __restrict__ maxrot,
                                                      const icpx_bnb_item* __restrict__ items, int nitems, int nsplit, double* __restrict__ partials,
                                                      float* __restrict__ mindis, const int block, const int split, double (*sm)[4], double (*sm8)[4]) {
    const int first = block * 8;
    icpx_bnb_item it;
    float tx[2], ty[2], tz[2];
    if (PARENTS) {
        it = items[block];
        it.w = it.w * 0.5f;                     // (float)(parent.w / 2.0): exact either way
        const float half = it.w / 2;
        tx[0] = it.x + half; tx[1] = (it.x + it.w) + half;
        ty[0] = it.y + half; ty[1] = (it.y + it.w) + half;
        tz[0] = it.z + half; tz[1] = (it.z + it.w) + half;
        if (it.level >= ICPX_MAXROTLEVEL) {    // a level beyond the table: NaN bounds, as in the one-item kernel
            if (threadIdx.x < 16) bnb_put<AGENT_OUT>(&partials[((size_t)(first + (threadIdx.x >> 1)) * nsplit + split) * 2 + (threadIdx.x & 1)], (double)NAN);
            return;
        }
    } else {
        const int cnt = nitems - first < 8 ? nitems - first : 8;
        bool family = cnt == 8;
        if (family) {
            const icpx_bnb_item a = items[first];
            family = a.level < ICPX_MAXROTLEVEL;
            for (int j = 1; j < 8; j++) {
                const icpx_bnb_item b = items[first + j];
                bool same = b.level == a.level && same_bits(b.w, a.w);
                for (int k = 0; k < 9; k++) same = same && same_bits(b.R[k], a.R[k]);
                same = same && same_bits(b.x, items[first + (j & 1)].x) && same_bits(b.y, items[first + (j & 2)].y) && same_bits(b.z, items[first + (j & 4)].z);
                family = family && same;
            }
        }
        if (!family) {
            for (int k = 0; k < cnt; k++) bnb_eval_item(e, data, n, maxrot, items, first + k, split, nsplit, partials, mindis, sm);
            return;
        }
        it = items[first];
        const float half = it.w / 2;
        tx[0] = it.x + half; tx[1] = items[first + 1].x + half;
        ty[0] = it.y + half; ty[1] = items[first + 2].y + half;
        tz[0] = it.z + half; tz[1] = items[first + 4].z + half;
    }
    const float maxTransDis = (float)(GOICP_SQRT3 / 2.0 * (double)it.w);
    const float* mr = it.level >= 0 ? maxrot + (size_t)it.level * n : nullptr;
    const float fdiv = (float)e.div;
    const float mxx = __fadd_rn(e.mnx, __fmul_rn(e.cell, fdiv)), mxy = __fadd_rn(e.mny, __fmul_rn(e.cell, fdiv)), mxz = __fadd_rn(e.mnz, __fmul_rn(e.cell, fdiv));
    const float inv = __frcp_rn(e.cell);
    const float eps_near = (float)(e.div + 1) * 1e-6f;
    const __amdgpu_buffer_rsrc_t grid = edt_grid_rsrc(e.sq);
    const __amdgpu_buffer_rsrc_t bricks = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(e.bricks), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t fbricks = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.fbricks), 0, -1, 0x00020000);
    const size_t per = (n + nsplit - 1) / nsplit;
    const size_t lo = (size_t)split * per, hi = lo + per < n ? lo + per : n;
    double ub[8], lb[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ub[j] = lb[j] = 0.0;
    // One point in two parts, so that the lookups of several points can be in flight before the first is added up.
    // lookup: the point rotated, the six cell coordinates of the family, the eight distances read (inside the box);
    // returns whether some coordinate of some child lies outside the box. finish: the overshoot added where there is
    // one, the rotation uncertainty subtracted, the terms of the sixteen sums added.
    auto lookup = [&](const float4& p, float (&d)[8], float (&vv)[6]) -> bool {
            const float rx = it.R[0] * p.x + it.R[1] * p.y + it.R[2] * p.z;
            const float ry = it.R[3] * p.x + it.R[4] * p.y + it.R[5] * p.z;
            const float rz = it.R[6] * p.x + it.R[7] * p.y + it.R[8] * p.z;
            EdtAxis ax[2], ay[2], az[2];
            const float vx[2] = {rx + tx[0], rx + tx[1]}, vy[2] = {ry + ty[0], ry + ty[1]}, vz[2] = {rz + tz[0], rz + tz[1]};
            vv[0] = vx[0]; vv[1] = vx[1]; vv[2] = vy[0]; vv[3] = vy[1]; vv[4] = vz[0]; vv[5] = vz[1];
            bool mv[6];
    #pragma unroll
            for (int s = 0; s < 2; s++) {
                ax[s] = edt_axis_begin_inside(vx[s], e.mnx, mxx, inv, eps_near, mv[3 * s]);
                ay[s] = edt_axis_begin_inside(vy[s], e.mny, mxy, inv, eps_near, mv[3 * s + 1]);
                az[s] = edt_axis_begin_inside(vz[s], e.mnz, mxz, inv, eps_near, mv[3 * s + 2]);
            }
            // some coordinate of some child outside the box (then the distances get the overshoot added, below)
            const bool moved = (mv[0] | mv[1] | mv[2]) | (mv[3] | mv[4] | mv[5]);
            // the division itself only in waves where some lane needs it (the empty asm keeps the compiler from turning the
            // branch into six unconditional divisions)
            if (__any(ax[0].near || ax[1].near || ay[0].near || ay[1].near || az[0].near || az[1].near)) {
                asm volatile("" ::: "memory");
    #pragma unroll
                for (int s = 0; s < 2; s++) { edt_axis_exact(ax[s], e.cell); edt_axis_exact(ay[s], e.cell); edt_axis_exact(az[s], e.cell); }
            }
            const unsigned cx[2] = {edt_axis_cell_inside(ax[0], e.div), edt_axis_cell_inside(ax[1], e.div)},
                           cy[2] = {edt_axis_cell_inside(ay[0], e.div), edt_axis_cell_inside(ay[1], e.div)},
                           cz[2] = {edt_axis_cell_inside(az[0], e.div), edt_axis_cell_inside(az[1], e.div)};
            float g[8];
            if (BRICKS) {
                // a voxel's place in the brick copy is a sum of one term per axis
                unsigned xo[2], xyo[4], zo[2];
    #pragma unroll
                for (int s = 0; s < 2; s++) {
                    xo[s] = (cx[s] >> 2) * (unsigned)(e.nb * e.nb * 64) + (cx[s] & 3) * 16;
                    zo[s] = (cz[s] >> 2) * 64 + (cz[s] & 3);
                }
    #pragma unroll
                for (int s = 0; s < 4; s++) xyo[s] = xo[s & 1] + (cy[s >> 1] >> 2) * (unsigned)(e.nb * 64) + (cy[s >> 1] & 3) * 4;
                if (BRICKS == 2) {
    #pragma unroll
                    for (int j = 0; j < 8; j++) d[j] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(fbricks, (int)((xyo[j & 3] + zo[j >> 2]) * 4u), 0, 0));
                } else {
                unsigned v[8];
    #pragma unroll
                for (int j = 0; j < 8; j++) v[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(bricks, (int)((xyo[j & 3] + zo[j >> 2]) * 2u), 0, 0) & 0xffffu;
                const unsigned top = max(max(max(v[0], v[1]), max(v[2], v[3])), max(max(v[4], v[5]), max(v[6], v[7])));
                if (__any(top == EDT_BRICK_ESCAPE)) {
    #pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (v[j] == EDT_BRICK_ESCAPE)
                            v[j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(grid, (int)(((cx[j & 1] * (unsigned)e.div + cy[j >> 1 & 1]) * (unsigned)e.zs + cz[j >> 2]) * 4u), 0, 0);
                }
    #pragma unroll
                for (int j = 0; j < 8; j++) g[j] = (float)v[j];
                }
            } else {
                unsigned xoff[2], xyoff[4];
    #pragma unroll
                for (int s = 0; s < 2; s++) xoff[s] = cx[s] * (unsigned)e.div;
    #pragma unroll
                for (int s = 0; s < 4; s++) xyoff[s] = (xoff[s & 1] + cy[s >> 1]) * (unsigned)e.zs;
    #pragma unroll
                for (int j = 0; j < 8; j++)
                    g[j] = (float)__builtin_amdgcn_raw_buffer_load_b32(grid, (int)((xyoff[j & 3] + cz[j >> 2]) * 4u), 0, 0);
            }
            if (BRICKS != 2) {
    #pragma unroll
                for (int j = 0; j < 8; j++) d[j] = __fmul_rn(edt_sqrt_of_int(g[j]), e.cell);
            }
        return moved;
    };
    auto finish = [&](const float4& p, const float mr_i, float (&d)[8], const float (&vv)[6], const bool some_outside) {
            if (some_outside) {
                const float vx[2] = {vv[0], vv[1]}, vy[2] = {vv[2], vv[3]}, vz[2] = {vv[4], vv[5]};
                // the overshoot per axis as LDT::Distance forms it: mn - v below the box, v - mx above it, else 0
                float ovx[2], ovy[2], ovz[2];
    #pragma unroll
                for (int s = 0; s < 2; s++) {
                    ovx[s] = fmaxf(fmaxf(__fsub_rn(e.mnx, vx[s]), __fsub_rn(vx[s], mxx)), 0.f);
                    ovy[s] = fmaxf(fmaxf(__fsub_rn(e.mny, vy[s]), __fsub_rn(vy[s], mxy)), 0.f);
                    ovz[s] = fmaxf(fmaxf(__fsub_rn(e.mnz, vz[s]), __fsub_rn(vz[s], mxz)), 0.f);
                }
    #pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float ox = ovx[j & 1], oy = ovy[j >> 1 & 1], oz = ovz[j >> 2];
                    if (ox != 0.f || oy != 0.f || oz != 0.f) {
                        const double dx = (double)ox, dy = (double)oy, dz = (double)oz;
                        const double over = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                        d[j] = (float)__dadd_rn((double)d[j], __dsqrt_rn(over));
                    }
                }
            }
            // (mr_i is 0 without a level: d - 0 is d; two children at a time, for the packed float instructions)
    #pragma unroll
            for (int j = 0; j < 8; j += 2) {
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f dj = v2f{d[j], d[j + 1]} - mr_i;
                dj = v2f{fmaxf(dj.x, 0.f), fmaxf(dj.y, 0.f)};
                if (mindis) { mindis[(size_t)(first + j) * n + __float_as_int(p.w)] = dj.x; mindis[(size_t)(first + j + 1) * n + __float_as_int(p.w)] = dj.y; }
                const v2f sq = dj * dj;
                ub[j] += (double)sq.x; ub[j + 1] += (double)sq.y;
                v2f dis = dj - maxTransDis;
                dis = v2f{fmaxf(dis.x, 0.f), fmaxf(dis.y, 0.f)};      // a term that is not positive adds +0, which leaves the sum as it is
                const v2f sq2 = dis * dis;
                lb[j] += (double)sq2.x; lb[j + 1] += (double)sq2.y;
            }
    };
    if (AHEAD <= 1) {
        size_t i = lo + threadIdx.x;
        float4 p_next = i < hi ? data[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float mr_next = mr && i < hi ? mr[i] : 0.f;
        for (; i < hi; i += blockDim.x) {
            const float4 p = p_next;
            const float mr_i = mr_next;
            if (i + blockDim.x < hi) { p_next = data[i + blockDim.x]; if (mr) mr_next = mr[i + blockDim.x]; }
            float d[8], vv[6];
            const bool moved = lookup(p, d, vv);
            finish(p, mr_i, d, vv, __any(moved));
        }
    } else {
        // Few workgroups in the launch (the last running searches of a step): the time is that of one thread's chain of
        // points, each a rotation, eight reads and the sums. AHEAD points are looked up together and added in order.
        constexpr int U = AHEAD > 1 ? AHEAD : 1;
        for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += (size_t)U * blockDim.x) {
            float4 p[U];
            float mr_i[U], d[U][8], vv[U][6];
            bool live[U], moved[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const size_t i = i0 + (size_t)u * blockDim.x;
                live[u] = i < hi;
                p[u] = live[u] ? data[i] : make_float4(0.f, 0.f, 0.f, 0.f);     // (a point that is not there is looked up at the origin and left out of the sums)
                mr_i[u] = mr && live[u] ? mr[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; u++) moved[u] = lookup(p[u], d[u], vv[u]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool some_outside = __any(moved[u] && live[u]);
                if (live[u]) finish(p[u], mr_i[u], d[u], vv[u], some_outside);
            }
        }
    }
    // The sixteen sums of the wave by one butterfly instead of sixteen: at each of the first four steps a lane hands the
    // half of its values that its partner keeps to the partner and adds what it receives to the half it keeps, so the
    // pairs added at every step are those of the plain butterfly over lanes l and l ^ o (whose result is the same bits
    // in every lane, addition being commutative), at a quarter of the instructions.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v[16];
#pragma unroll
    for (int j = 0; j < 8; j++) { v[j] = ub[j]; v[8 + j] = lb[j]; }
    bnb_fold<8>(v, lane, 32);
    bnb_fold<4>(v, lane, 16);
    bnb_fold<2>(v, lane, 8);
    bnb_fold<1>(v, lane, 4);
    v[0] += __shfl_xor(v[0], 2);
    v[0] += __shfl_xor(v[0], 1);
    if ((lane & 3) == 0) {
        const int which = lane >> 2;                 // 0 .. 7: ub of child `which`, 8 .. 15: lb of child `which - 8`
        sm8[which < 8 ? 2 * which : 2 * (which - 8) + 1][wave] = v[0];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const double v = (sm8[threadIdx.x][0] + sm8[threadIdx.x][1]) + (sm8[threadIdx.x][2] + sm8[threadIdx.x][3]);
        bnb_put<AGENT_OUT>(&partials[((size_t)(first + (threadIdx.x >> 1)) * nsplit + split) * 2 + (threadIdx.x & 1)], v);
    }
}

template <int BRICKS, bool PARENTS>
__global__ __launch_bounds__(256) void k_bnb_eval8(EdtView e, const float4* __restrict__ data, size_t n, const float* __restrict__ maxrot,
                                                   const icpx_bnb_item* __restrict__ items, int nitems, int nsplit, double* __restrict__ partials,
                                                   float* __restrict__ mindis) {
    __shared__ double sm[2][4];
    __shared__ double sm8[16][4];
    bnb_eval8_body<BRICKS, PARENTS>(e, data, n, maxrot, items, nitems, nsplit, partials, mindis, (int)blockIdx.x, (int)blockIdx.y, sm, sm8);
}

// The eight children of every parent cube as items of their own, for batches too small to fill the chip eight items
// per workgroup.
__global__ void k_bnb_expand(const icpx_bnb_item* __restrict__ parents, int nparents, icpx_bnb_item* __restrict__ items) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nparents * 8) return;
    icpx_bnb_item it = parents[i >> 3];
    const int j = i & 7;
    it.w = it.w * 0.5f;
    it.x = it.x + (float)(j & 1) * it.w;
    it.y = it.y + (float)(j >> 1 & 1) * it.w;
    it.z = it.z + (float)(j >> 2 & 1) * it.w;
    items[i] = it;
}

// out: [ub_0 .. ub_{total-1}, lb_0 .. lb_{total-1}]. The partials belong to items first .. first + count - 1 of the batch; the
// other entries (other ranks' shares of a batch that is summed over the ranks afterwards) are set to zero.
__global__ void k_bnb_reduce(const double* __restrict__ partials, int total, int first, int count, int nsplit, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    double ub = 0.0, lb = 0.0;
    const int k = i - first;
    if (k >= 0 && k < count)
        for (int s = 0; s < nsplit; s++) { ub += partials[((size_t)k * nsplit + s) * 2]; lb += partials[((size_t)k * nsplit + s) * 2 + 1]; }
    out[i] = ub;
    out[total + i] = lb;
}

struct Mat34 { double m[12]; };

__global__ void k_goicp_residuals(EdtView e, const float4* __restrict__ data, size_t n, Mat34 T, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 s = data[i];
    double p[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double a = 0.0;
        a += T.m[4 * r + 0] * (double)s.x;
        a += T.m[4 * r + 1] * (double)s.y;
        a += T.m[4 * r + 2] * (double)s.z;
        a += T.m[4 * r + 3];
        p[r] = a;
    }
    out[__float_as_int(s.w)] = edt_distance(e.sq, e.div, e.zs, e.mnx, e.mny, e.mnz, e.cell, (float)p[0], (float)p[1], (float)p[2]);
}

EdtView edt_view(const icpx_ctx* ctx) {
    EdtView e;
    e.sq = ctx->edt_sq.as<int>(); e.div = (int)ctx->div; e.zs = (int)ctx->edt_zs;
    e.mnx = ctx->edt_min[0]; e.mny = ctx->edt_min[1]; e.mnz = ctx->edt_min[2]; e.cell = ctx->edt_cell;
    e.bricks = ctx->edt_bricks_ready ? ctx->edt_bricks.as<unsigned short>() : nullptr;
    e.fbricks = ctx->edt_bricks_ready && ctx->edt_fbricks_ready ? ctx->edt_fbricks.as<float>() : nullptr;
    e.nb = ((int)ctx->div + 3) / 4;
    return e;
}

}  // namespace

extern "C" int icpx_goicp_set_data(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where) {
    if (!ctx) return ICPX_EINVAL;
    if (!xyz || n == 0) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_goicp_set_data: empty data cloud");
    if (stride_bytes < 12 || stride_bytes % 4) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_goicp_set_data: stride must be a multiple of 4 and >= 12");
    if (n > 0x7fffffffu) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_goicp_set_data: more than 2^31-1 points");
    ICPX_TRY(icpx_bind(ctx));
    ctx->nd = 0;
    ICPX_TRY(icpx_reserve(ctx, ctx->data_pts, n * sizeof(float4)));
    ICPX_TRY(icpx_reserve(ctx, ctx->maxrot, (size_t)ICPX_MAXROTLEVEL * n * sizeof(float)));
    ICPX_TRY(icpx_stage_points4(ctx, xyz, n, stride_bytes, where, ctx->data_pts.as<float4>()));
    ICPX_TRY(icpx_morton_sort4(ctx, ctx->data_pts.as<float4>(), n));
    RotConsts rc;
    // InitRot.w is a float member (RotNode, src/GoIcp.cpp:26): the 2*PI assigned at :88 is rounded to float and :472
    // widens that float, not the double product
    const double initRotW = (double)(float)(2 * GOICP_PI);
    for (int l = 0; l < ICPX_MAXROTLEVEL; l++) {
        const double sigma = initRotW / std::pow(2.0, l) / 2.0;
        double tmp = sigma * GOICP_SQRT3;
        if (tmp > GOICP_PI) tmp = GOICP_PI;
        rc.c[l] = 2 * std::sin(tmp / 2);
    }
    hipLaunchKernelGGL(k_maxrotdist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->data_pts.as<float4>(), n, rc,
                       ctx->maxrot.as<float>());
    ICPX_TRY(icpx_launch_check(ctx, "k_maxrotdist"));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nd = n;
    return ICPX_OK;
}

__global__ void k_unsort_rows(const float* __restrict__ src, const float4* __restrict__ data, size_t n, int rows, float* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int o = __float_as_int(data[i].w);
    for (int r = 0; r < rows; r++) dst[(size_t)r * n + o] = src[(size_t)r * n + i];
}

extern "C" int icpx_goicp_read_maxrotdist(icpx_ctx* ctx, float* out) {
    if (!ctx) return ICPX_EINVAL;
    if (ctx->nd == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_goicp_read_maxrotdist: no data cloud (call icpx_goicp_set_data first)");
    if (!out) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_goicp_read_maxrotdist: out is NULL");
    ICPX_TRY(icpx_bind(ctx));
    const size_t n = ctx->nd, bytes = (size_t)ICPX_MAXROTLEVEL * n * sizeof(float);
    ICPX_TRY(icpx_reserve(ctx, ctx->bnb_mindis, bytes));
    hipLaunchKernelGGL(k_unsort_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->maxrot.as<float>(),
                       ctx->data_pts.as<float4>(), n, ICPX_MAXROTLEVEL, ctx->bnb_mindis.as<float>());
    ICPX_TRY(icpx_launch_check(ctx, "k_unsort_rows"));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(out, ctx->bnb_mindis.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

// Grow-only pinned host buffer: copies through it are truly asynchronous and take one DMA each, where a
// pageable source or destination is staged by the runtime in several steps.
#ifndef ICPX_BNB_SLICES
#define ICPX_BNB_SLICES 16
#endif

// Slices per item: a function of the cloud alone, never of the batch, so that the bounds of a cube are the same
// bits whichever cubes it shares a launch with (the partial sums are added in slice order).
static int bnb_slices(size_t n) {
    const int nsplit = (int)std::min<size_t>((n + 1023) / 1024, ICPX_BNB_SLICES);
    return nsplit < 1 ? 1 : nsplit;
}

// Partial sums of icpx_bnb_eval_noalloc: {ub, lb} per (item, slice), ICPX_NOALLOC_MAX_ITEMS x ICPX_BNB_SLICES x 2
// doubles = 1 MiB, part of the code object, so every GPU the library runs on holds its own copy at its own
// address until the process ends. In host code the array's name is only a handle for hipGetSymbolAddress: the
// address kernels and copies use is the one icpx_bnb_scratch_init stores in the context.
static __device__ __align__(256) double scratch[(size_t)ICPX_NOALLOC_MAX_ITEMS * ICPX_BNB_SLICES * 2];

// Called by icpx_create with the context's device current.
int icpx_bnb_scratch_init(icpx_ctx* ctx) {
    void* ptr = nullptr;
    ICPX_CHECK_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) ICPX_FAIL(ctx, ICPX_EHIP, "hipGetSymbolAddress returned no address for the bound scratch array");
    ctx->bnb_scratch = static_cast<double*>(ptr);
    return ICPX_OK;
}

static int bnb_pin_reserve(icpx_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->bnb_pin_cap) return ICPX_OK;
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->bnb_pin) { ICPX_CHECK_HIP(ctx, hipHostFree(ctx->bnb_pin)); ctx->bnb_pin = nullptr; ctx->bnb_pin_cap = 0; }
    const size_t want = std::max<size_t>(bytes * 2, 1 << 16);
    ICPX_CHECK_HIP(ctx, hipHostMalloc(&ctx->bnb_pin, want, hipHostMallocDefault));
    ctx->bnb_pin_cap = want;
    return ICPX_OK;
}

// The bound kernel over `nitems` items in device memory: per (item, slice) partial sums into `partials`.
// Makes the brick copy of the current grid if there is none yet (any EDT build discards it). Allocates on first use.
static int bnb_bricks_ensure(icpx_ctx* ctx) {
    if (ctx->edt_bricks_ready) return ICPX_OK;
    const size_t nb = ((size_t)ctx->div + 3) / 4, total = nb * nb * nb * 64;
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_bricks, total * sizeof(unsigned short)));
    hipLaunchKernelGGL(k_edt_bricks, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const int*)ctx->edt_sq.as<int>(), (int)ctx->div, (int)ctx->edt_zs, (int)nb,
                       ctx->edt_bricks.as<unsigned short>());
    ICPX_TRY(icpx_launch_check(ctx, "k_edt_bricks"));
    // the float bricks take 256 bytes per 64 voxels: up to a gigabyte of them (grids of some 640 cells a side)
    ctx->edt_fbricks_ready = false;
    if (ctx->bnb_bricks == 2 && total * sizeof(float) <= ((size_t)1 << 30)) {
        ICPX_TRY(icpx_reserve(ctx, ctx->edt_fbricks, total * sizeof(float)));
        hipLaunchKernelGGL(k_edt_fbricks, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const int*)ctx->edt_sq.as<int>(), (int)ctx->div, (int)ctx->edt_zs, (int)nb,
                           ctx->edt_cell, ctx->edt_fbricks.as<float>());
        ICPX_TRY(icpx_launch_check(ctx, "k_edt_fbricks"));
        ctx->edt_fbricks_ready = true;
    }
    ctx->edt_bricks_ready = true;
    return ICPX_OK;
}

// The bound kernel over `nitems` items in device memory: per (item, slice) partial sums into `partials`. The kernel
// that takes eight items per workgroup reads the brick copy of the grid when there is one.
static void bnb_launch(icpx_ctx* ctx, const icpx_bnb_item* items_dev, size_t nitems, int nsplit, double* partials, float* mindis_dev) {
    const size_t n = ctx->nd;
    auto eight = k_bnb_eval8<0, false>;
    if (ctx->edt_bricks_ready) eight = ctx->edt_fbricks_ready ? k_bnb_eval8<2, false> : k_bnb_eval8<1, false>;
    if (nitems >= ctx->bnb_group_min && nitems >= 8)
        hipLaunchKernelGGL(eight, dim3((unsigned)((nitems + 7) / 8), (unsigned)nsplit), dim3(256), 0,
                           ctx->stream, edt_view(ctx), ctx->data_pts.as<float4>(), n, ctx->maxrot.as<float>(), items_dev, (int)nitems, nsplit, partials, mindis_dev);
    else
        hipLaunchKernelGGL(k_bnb_eval, dim3((unsigned)nitems, (unsigned)nsplit), dim3(256), 0, ctx->stream, edt_view(ctx),
                           ctx->data_pts.as<float4>(), n, ctx->maxrot.as<float>(), items_dev, nsplit, partials, mindis_dev);
}

// Evaluates items[0 .. nitems) — entries first .. first + nitems - 1 of a batch of `total` — into out_dev (2 * total doubles).
static int bnb_enqueue(icpx_ctx* ctx, const icpx_bnb_item* items, size_t nitems, double* out_dev, float* mindis_dev, size_t total, size_t first) {
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_eval: no EDT has been built");
    if (ctx->nd == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_eval: no data cloud (call icpx_goicp_set_data first)");
    if (!items && nitems) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_eval: items is NULL");
    if (total > 65535u * 16u) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_bnb_eval: more than 1048560 items in one batch");
    for (size_t i = 0; i < nitems; i++)
        if (items[i].level >= ICPX_MAXROTLEVEL) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_eval: item %zu has level %d >= %d", i, items[i].level, ICPX_MAXROTLEVEL);
    const size_t n = ctx->nd;
    const int nsplit = bnb_slices(n);
    ICPX_TRY(icpx_reserve(ctx, ctx->bnb_items, std::max<size_t>(nitems, 1) * sizeof(icpx_bnb_item)));
    ICPX_TRY(icpx_reserve(ctx, ctx->bnb_partials, std::max<size_t>(nitems, 1) * nsplit * 2 * sizeof(double)));
    // The item list goes through the pinned buffer (front part; the results come back into the part behind
    // it), so the caller may reuse `items` as soon as this function returns. A previous batch that is still
    // in flight on the stream would read the same staging area: batches are serialised here.
    const size_t items_bytes = nitems * sizeof(icpx_bnb_item);
    ICPX_TRY(bnb_pin_reserve(ctx, items_bytes + 2 * total * sizeof(double)));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nitems) {
        if (ctx->bnb_bricks && nitems >= ctx->bnb_group_min && nitems >= 8) ICPX_TRY(bnb_bricks_ensure(ctx));
        std::memcpy(ctx->bnb_pin, items, items_bytes);
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->bnb_items.p, ctx->bnb_pin, items_bytes, hipMemcpyHostToDevice, ctx->stream));
        bnb_launch(ctx, ctx->bnb_items.as<icpx_bnb_item>(), nitems, nsplit, ctx->bnb_partials.as<double>(), mindis_dev);
    }
    hipLaunchKernelGGL(k_bnb_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->bnb_partials.as<double>(),
                       (int)total, (int)first, (int)nitems, nsplit, out_dev);
    return icpx_launch_check(ctx, "k_bnb_eval");
}

extern "C" int icpx_bnb_eval(icpx_ctx* ctx, const icpx_bnb_item* items, size_t nitems, double* ub, double* lb, float* mindis_opt) {
    if (!ctx) return ICPX_EINVAL;
    if (nitems == 0) return ICPX_OK;
    if (!ub || !lb) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_eval: ub / lb is NULL");
    ICPX_TRY(icpx_bind(ctx));
    float* md = nullptr;
    if (mindis_opt) {
        ICPX_TRY(icpx_reserve(ctx, ctx->bnb_mindis, nitems * ctx->nd * sizeof(float)));
        md = ctx->bnb_mindis.as<float>();
    }
    // The reduction kernel writes the bounds into the pinned staging buffer itself (behind the items): one copy
    // command fewer per batch. The buffer is sized here so that the address is known before the launch.
    ICPX_TRY(bnb_pin_reserve(ctx, nitems * sizeof(icpx_bnb_item) + 2 * nitems * sizeof(double)));
    double* res = reinterpret_cast<double*>(static_cast<char*>(ctx->bnb_pin) + nitems * sizeof(icpx_bnb_item));
    ICPX_TRY(bnb_enqueue(ctx, items, nitems, res, md, nitems, 0));
    if (mindis_opt)
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(mindis_opt, md, nitems * ctx->nd * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(ub, res, nitems * sizeof(double));
    std::memcpy(lb, res + nitems, nitems * sizeof(double));
    return ICPX_OK;
}

extern "C" int icpx_bnb_eval_async(icpx_ctx* ctx, const icpx_bnb_item* items, size_t nitems, size_t first, size_t count, double* out_dev) {
    if (!ctx) return ICPX_EINVAL;
    if (nitems == 0) return ICPX_OK;
    if (!out_dev) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_eval_async: out_dev is NULL");
    if (first > nitems || count > nitems - first) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_eval_async: the share [first, first + count) lies outside the batch");
    if (!items) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_eval_async: items is NULL");
    ICPX_TRY(icpx_bind(ctx));
    return bnb_enqueue(ctx, items + first, count, out_dev, nullptr, nitems, first);
}

// ---- parent cubes, two batches in flight ------------------------------------------------------------------------
// A slot owns what one batch in flight needs: pinned host staging (the parents going up, the bounds coming down),
// the records, partial sums and bounds on the device, and the event that says its bounds have arrived.
static int bnb_slot_reserve(icpx_ctx* ctx, BnbSlot& sl, size_t nparents, size_t count) {
    const size_t in_bytes = std::max<size_t>(count, 1) * sizeof(icpx_bnb_item), out_bytes = 16 * nparents * sizeof(double);
    if (in_bytes + out_bytes > sl.pin_cap) {
        if (sl.pin) { ICPX_CHECK_HIP(ctx, hipHostFree(sl.pin)); sl.pin = nullptr; sl.pin_cap = 0; }
        const size_t want = std::max<size_t>((in_bytes + out_bytes) * 2, 1 << 16);
        ICPX_CHECK_HIP(ctx, hipHostMalloc(&sl.pin, want, hipHostMallocDefault));
        sl.pin_cap = want;
    }
    if (!sl.done) ICPX_CHECK_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    ICPX_TRY(icpx_reserve(ctx, sl.parents, std::max<size_t>(count, 1) * sizeof(icpx_bnb_item)));
    ICPX_TRY(icpx_reserve(ctx, sl.items, std::max<size_t>(count, 1) * 8 * sizeof(icpx_bnb_item)));
    ICPX_TRY(icpx_reserve(ctx, sl.partials, std::max<size_t>(count, 1) * 8 * ICPX_BNB_SLICES * 2 * sizeof(double)));
    return ICPX_OK;
}

extern "C" int icpx_bnb_submit(icpx_ctx* ctx, int slot, const icpx_bnb_item* parents, size_t nparents, size_t first, size_t count, double* out_dev) {
    if (!ctx) return ICPX_EINVAL;
    if (slot < 0 || slot >= ICPX_BNB_SLOTS) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_submit: slot must be 0 .. %d", ICPX_BNB_SLOTS - 1);
    if (!parents || nparents == 0) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_submit: empty batch");
    if (first > nparents || count > nparents - first) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_submit: the share [first, first + count) lies outside the batch");
    if (nparents > 65535u * 2u) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_bnb_submit: more than 131070 parent cubes in one batch");
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_submit: no EDT has been built");
    if (ctx->nd == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_submit: no data cloud (call icpx_goicp_set_data first)");
    for (size_t i = first; i < first + count; i++)
        if (parents[i].level >= ICPX_MAXROTLEVEL) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_submit: parent %zu has level %d >= %d", i, parents[i].level, ICPX_MAXROTLEVEL);
    ICPX_TRY(icpx_bind(ctx));
    BnbSlot& sl = ctx->bnb_slot[slot];
    // the slot's previous batch has been collected, or is abandoned here: its staging area is about to be rewritten
    if (sl.busy) { ICPX_CHECK_HIP(ctx, hipEventSynchronize(sl.done)); sl.busy = false; }
    ICPX_TRY(bnb_slot_reserve(ctx, sl, nparents, count));
    const size_t n = ctx->nd;
    const int nsplit = bnb_slices(n);
    const size_t nchild = 8 * count, total = 8 * nparents;
    const bool eight = nchild >= ctx->bnb_group_min;
    if (count && ctx->bnb_bricks && eight) ICPX_TRY(bnb_bricks_ensure(ctx));
    // Both slots share the context's stream: a stream of its own per slot, so that one slot's copies could overlap the
    // other's kernels, was tried and gained nothing (the loop is limited by the host side of the search).
    hipStream_t st = ctx->stream;
    if (count) {
        std::memcpy(sl.pin, parents + first, count * sizeof(icpx_bnb_item));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(sl.parents.p, sl.pin, count * sizeof(icpx_bnb_item), hipMemcpyHostToDevice, st));
        if (eight) {
            auto kernel = k_bnb_eval8<0, true>;
            if (ctx->edt_bricks_ready) kernel = ctx->edt_fbricks_ready ? k_bnb_eval8<2, true> : k_bnb_eval8<1, true>;
            hipLaunchKernelGGL(kernel, dim3((unsigned)count, (unsigned)nsplit), dim3(256), 0, st,
                               edt_view(ctx), ctx->data_pts.as<float4>(), n, ctx->maxrot.as<float>(), sl.parents.as<icpx_bnb_item>(), (int)nchild, nsplit,
                               sl.partials.as<double>(), (float*)nullptr);
        } else {
            hipLaunchKernelGGL(k_bnb_expand, dim3((unsigned)((nchild + 255) / 256)), dim3(256), 0, st, sl.parents.as<icpx_bnb_item>(), (int)count,
                               sl.items.as<icpx_bnb_item>());
            hipLaunchKernelGGL(k_bnb_eval, dim3((unsigned)nchild, (unsigned)nsplit), dim3(256), 0, st, edt_view(ctx), ctx->data_pts.as<float4>(), n,
                               ctx->maxrot.as<float>(), sl.items.as<icpx_bnb_item>(), nsplit, sl.partials.as<double>(), (float*)nullptr);
        }
    }
    // bounds for the host: the reduction kernel writes them into the pinned staging area itself (behind the records)
    sl.to_host = out_dev == nullptr;
    sl.result = reinterpret_cast<double*>(static_cast<char*>(sl.pin) + std::max<size_t>(count, 1) * sizeof(icpx_bnb_item));
    hipLaunchKernelGGL(k_bnb_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sl.partials.as<double>(), (int)total, (int)(8 * first),
                       (int)nchild, nsplit, out_dev ? out_dev : sl.result);
    ICPX_TRY(icpx_launch_check(ctx, "icpx_bnb_submit"));
    sl.nparents = nparents;
    ICPX_CHECK_HIP(ctx, hipEventRecord(sl.done, st));
    sl.busy = true;
    return ICPX_OK;
}

extern "C" int icpx_bnb_fetch(icpx_ctx* ctx, int slot, const double* out_dev) {
    if (!ctx) return ICPX_EINVAL;
    if (slot < 0 || slot >= ICPX_BNB_SLOTS) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_fetch: slot must be 0 .. %d", ICPX_BNB_SLOTS - 1);
    BnbSlot& sl = ctx->bnb_slot[slot];
    if (!sl.busy || sl.to_host) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_fetch: slot %d holds no batch that was submitted with a device buffer", slot);
    if (!out_dev) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_fetch: out_dev is NULL");
    ICPX_TRY(icpx_bind(ctx));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(sl.result, out_dev, 16 * sl.nparents * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipEventRecord(sl.done, ctx->stream));
    sl.to_host = true;
    return ICPX_OK;
}

extern "C" int icpx_bnb_collect(icpx_ctx* ctx, int slot, double* ub, double* lb) {
    if (!ctx) return ICPX_EINVAL;
    if (slot < 0 || slot >= ICPX_BNB_SLOTS) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_collect: slot must be 0 .. %d", ICPX_BNB_SLOTS - 1);
    BnbSlot& sl = ctx->bnb_slot[slot];
    if (!sl.busy) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_collect: nothing was submitted to slot %d", slot);
    ICPX_TRY(icpx_bind(ctx));
    ICPX_CHECK_HIP(ctx, hipEventSynchronize(sl.done));
    sl.busy = false;
    if (!sl.to_host) return ICPX_OK;      // the bounds are in the caller's device buffer
    if (!ub || !lb) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_collect: ub / lb is NULL");
    std::memcpy(ub, sl.result, 8 * sl.nparents * sizeof(double));
    std::memcpy(lb, sl.result + 8 * sl.nparents, 8 * sl.nparents * sizeof(double));
    return ICPX_OK;
}

extern "C" int icpx_download(icpx_ctx* ctx, const void* dev, void* host, size_t bytes) {
    if (!ctx) return ICPX_EINVAL;
    if (bytes == 0) return ICPX_OK;
    if (!dev || !host) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_download: NULL pointer");
    ICPX_TRY(icpx_bind(ctx));
    // through the context's pinned staging area: one DMA, where a pageable destination is staged by the runtime in pieces
    ICPX_TRY(bnb_pin_reserve(ctx, bytes));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->bnb_pin, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(host, ctx->bnb_pin, bytes);
    return ICPX_OK;
}

extern "C" int icpx_upload(icpx_ctx* ctx, void* dev, const void* host, size_t bytes) {
    if (!ctx) return ICPX_EINVAL;
    if (bytes == 0) return ICPX_OK;
    if (!dev || !host) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_upload: NULL pointer");
    ICPX_TRY(icpx_bind(ctx));
    ICPX_TRY(bnb_pin_reserve(ctx, bytes));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // an earlier transfer may still be reading the staging area
    std::memcpy(ctx->bnb_pin, host, bytes);
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(dev, ctx->bnb_pin, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ICPX_OK;
}

extern "C" int icpx_bnb_eval_noalloc(icpx_ctx* ctx, const icpx_bnb_item* items_dev, size_t nitems, double* out_dev) {
    if (!ctx) return ICPX_EINVAL;
    if (nitems == 0) return ICPX_OK;
    if (!items_dev || !out_dev) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_eval_noalloc: NULL buffer");
    if (nitems > ICPX_NOALLOC_MAX_ITEMS)
        ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_bnb_eval_noalloc: %zu items exceed the fixed scratch area (at most %d per call)", nitems, ICPX_NOALLOC_MAX_ITEMS);
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_eval_noalloc: no EDT has been built");
    if (ctx->nd == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_eval_noalloc: no data cloud (call icpx_goicp_set_data first)");
    if (!ctx->bnb_scratch) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_eval_noalloc: the scratch area was not resolved at icpx_create");
    ICPX_TRY(icpx_bind(ctx));
    const size_t n = ctx->nd;
    const int nsplit = bnb_slices(n);
    // Every {ub, lb} slot that k_bnb_reduce reads is written by k_bnb_eval of this same call, so nothing left in the
    // area by an earlier (possibly larger) batch is ever read and no clearing is needed.
    double* part = ctx->bnb_scratch;
    bnb_launch(ctx, items_dev, nitems, nsplit, part, nullptr);
    hipLaunchKernelGGL(k_bnb_reduce, dim3((unsigned)((nitems + 255) / 256)), dim3(256), 0, ctx->stream, part, (int)nitems, 0, (int)nitems, nsplit, out_dev);
    return icpx_launch_check(ctx, "k_bnb_eval (noalloc)");
}

// ---- inner (translation) searches run on the device ---------------------------------------------------------------
// GoIcp::innerBnB (src/GoIcp.cpp:353-457) keeps a priority queue of translation cubes: take the cube with the lowest
// lower bound, bound its eight children, record a better upper bound, queue the children that can still hold a better
// pose, stop when a child's lower bound comes within the threshold of the best error or the queue runs dry. A rotation
// cube needs two such searches per child cube and they are independent of each other, so many run side by side; a step
// of all of them is one launch (k_search_eval_step): the bound kernel over the list of cubes taken, one per running
// search, and, by the workgroup that finishes a cube's last slice, the step of that search: the slices summed, the eight
// bounds taken as the loop above takes them, the search's queue moved, the cube taken next put on the following step's
// list. The host only enqueues steps and looks at the length of the list now and then.
//
// The queue is the binary heap of std::priority_queue<TransNode> (operator<: lb greater, src/GoIcp.cpp:35-42), moved by the push_heap /
// pop_heap steps of the GNU C++ library, because cubes of equal lower bound (the many with lb = 0 near the root) leave
// the queue in the order those steps give them, and that order decides which cube is expanded next.
#define ICPX_SEARCH_HEAP_CAP 1024      // cubes a search can hold in its queue
#define ICPX_SEARCH_MAX_STEPS 4096     // steps of one icpx_bnb_search_run call

struct SearchState {
    float R[9];
    int level;
    float opt;                // best error so far (optErrorT)
    int have_best;
    float best[4];            // the cube that gave it
    float parent[4];          // the cube whose children are being bounded
    int count;                // cubes in the queue
    int done, pops, status;
};

__global__ void k_search_start(const icpx_bnb_search* __restrict__ in, int n, SearchState* __restrict__ st, int* __restrict__ active,
                               icpx_bnb_item* __restrict__ parents, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) cnt[0] = n;
    if (i >= n) return;
    const icpx_bnb_search q = in[i];
    SearchState s;
    icpx_bnb_item it;
    for (int k = 0; k < 9; k++) { s.R[k] = q.R[k]; it.R[k] = q.R[k]; }
    s.level = q.level; it.level = q.level;
    s.opt = q.opt_error;
    s.have_best = 0;
    for (int k = 0; k < 4; k++) { s.best[k] = 0.f; s.parent[k] = q.cube[k]; }
    it.x = q.cube[0]; it.y = q.cube[1]; it.z = q.cube[2]; it.w = q.cube[3];
    s.count = 0; s.done = 0; s.pops = 0; s.status = 0;
    st[i] = s;
    active[i] = i;
    parents[i] = it;
}

// One step of one running search, by the workgroup that holds the last of its slices: the bounds of the eight children
// of active search a are summed over the slices and taken as the loop of GoIcp::innerBnB takes them (src/GoIcp.cpp:432-452),
// the search's queue moves, and the cube taken next goes into the list of the following step.
// Called by all threads of the workgroup; h_cube / h_lb / s_bound / s_n: its scratch in LDS. The partial sums were
// written by other workgroups of this launch, on any XCD: they are read past the caches (agent-scope loads) after the
// fence that follows the arrival count.
static __device__ __forceinline__ void search_step_one(const int a, SearchState* __restrict__ states, float4* __restrict__ heap_cube, float* __restrict__ heap_lb,
                                                       const double* partials, int nsplit, const int* __restrict__ active_in,
                                                       int* __restrict__ active_out, int* __restrict__ cnt_out, icpx_bnb_item* __restrict__ parents_out,
                                                       float sse_thresh, float4* h_cube, float* h_lb, float* s_bound, int* s_n) {
    const int tid = (int)threadIdx.x, nthreads = (int)blockDim.x;
    const int sid = active_in[a];
    if (tid < 16) {
        // child tid >> 1, ub (even) or lb (odd): the slices in order from zero, as k_bnb_reduce adds them
        double v = 0.0;
        for (int sl = 0; sl < nsplit; sl++)
            v += __hip_atomic_load(&partials[((size_t)(a * 8 + (tid >> 1)) * nsplit + sl) * 2 + (tid & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_bound[tid] = (float)v;
    }
    const int n0 = states[sid].count;
    const size_t hbase = (size_t)sid * ICPX_SEARCH_HEAP_CAP;
    for (int k = tid; k < n0; k += nthreads) { h_cube[k] = heap_cube[hbase + k]; h_lb[k] = heap_lb[hbase + k]; }
    __syncthreads();
    if (tid == 0) {
        SearchState s = states[sid];
        int n = n0;
        s.pops++;
        const float w2 = s.parent[3] * 0.5f;              // (float)(w / 2.0), exact
        for (int j = 0; j < 8 && !s.done; j++) {
            const float cx = s.parent[0] + (float)(j & 1) * w2, cy = s.parent[1] + (float)(j >> 1 & 1) * w2, cz = s.parent[2] + (float)(j >> 2 & 1) * w2;
            const float u = s_bound[2 * j], l = s_bound[2 * j + 1];
            if (u < s.opt) { s.opt = u; s.best[0] = cx; s.best[1] = cy; s.best[2] = cz; s.best[3] = w2; s.have_best = 1; }
            if (l >= s.opt) continue;
            if ((double)s.opt - (double)l < (double)sse_thresh) { s.done = 1; continue; }
            if (n >= ICPX_SEARCH_HEAP_CAP) { s.done = 1; s.status = 1; continue; }
            // push_heap: the new cube moves up past every parent with a greater lower bound
            int hole = n++;
            while (hole > 0) {
                const int par = (hole - 1) / 2;
                if (!(h_lb[par] > l)) break;
                h_cube[hole] = h_cube[par]; h_lb[hole] = h_lb[par];
                hole = par;
            }
            h_cube[hole] = make_float4(cx, cy, cz, w2); h_lb[hole] = l;
        }
        if (!s.done && n == 0) s.done = 1;
        if (!s.done) {
            // pop_heap: the top leaves, the hole moves down along the children with the smaller lower bound (the left
            // one when equal), and the last cube is pushed up from there
            const float4 top = h_cube[0];
            n--;
            if (n > 0) {
                const float4 vc = h_cube[n];
                const float vl = h_lb[n];
                const int len = n;
                int hole = 0, child = 0;
                while (child < (len - 1) / 2) {
                    child = 2 * (child + 1);
                    if (h_lb[child] > h_lb[child - 1]) child--;
                    h_cube[hole] = h_cube[child]; h_lb[hole] = h_lb[child];
                    hole = child;
                }
                if ((len & 1) == 0 && child == (len - 2) / 2) {
                    child = 2 * (child + 1);
                    h_cube[hole] = h_cube[child - 1]; h_lb[hole] = h_lb[child - 1];
                    hole = child - 1;
                }
                while (hole > 0) {
                    const int par = (hole - 1) / 2;
                    if (!(h_lb[par] > vl)) break;
                    h_cube[hole] = h_cube[par]; h_lb[hole] = h_lb[par];
                    hole = par;
                }
                h_cube[hole] = vc; h_lb[hole] = vl;
            }
            s.parent[0] = top.x; s.parent[1] = top.y; s.parent[2] = top.z; s.parent[3] = top.w;
            const int slot = atomicAdd(cnt_out, 1);
            active_out[slot] = sid;
            icpx_bnb_item it;
            for (int k = 0; k < 9; k++) it.R[k] = s.R[k];
            it.x = top.x; it.y = top.y; it.z = top.z; it.w = top.w; it.level = s.level;
            parents_out[slot] = it;
        }
        s.count = n;
        states[sid] = s;
        *s_n = s.done ? 0 : n;           // a finished search's queue is not read again
    }
    __syncthreads();
    const int n1 = *s_n;
    for (int k = tid; k < n1; k += nthreads) { heap_cube[hbase + k] = h_cube[k]; heap_lb[hbase + k] = h_lb[k]; }
    __syncthreads();
}

// A step of all running searches in one launch. The list of cubes to bound (one per running search, *cnt_in of them,
// written by the previous step) is worked off by workgroups (x, slice): x, x + gridDim.x, ... so that a grid sized for
// an earlier, longer list does no harm. The workgroup that finishes the last slice of a cube (an arrival count per
// list entry, back at zero afterwards) goes on with that search's step.
template <int BRICKS, int AHEAD>
__global__ __launch_bounds__(256) void k_search_eval_step(EdtView e, const float4* __restrict__ data, size_t n, const float* __restrict__ maxrot,
                                                          const icpx_bnb_item* __restrict__ parents_in, const int* __restrict__ cnt_in, int nsplit,
                                                          double* partials, int* __restrict__ arrive, SearchState* __restrict__ states,
                                                          float4* __restrict__ heap_cube, float* __restrict__ heap_lb, const int* __restrict__ active_in,
                                                          int* __restrict__ active_out, int* __restrict__ cnt_out, icpx_bnb_item* __restrict__ parents_out,
                                                          float sse_thresh) {
    __shared__ double sm[2][4];
    __shared__ double sm8[16][4];
    __shared__ float4 h_cube[ICPX_SEARCH_HEAP_CAP];
    __shared__ float h_lb[ICPX_SEARCH_HEAP_CAP];
    __shared__ float s_bound[16];
    __shared__ int s_n, s_last;
    const int cnt = *cnt_in;
    for (int a = (int)blockIdx.x; a < cnt; a += (int)gridDim.x) {
        bnb_eval8_body<BRICKS, true, AHEAD, true>(e, data, n, maxrot, parents_in, 8 * cnt, nsplit, partials, nullptr, a, (int)blockIdx.y, sm, sm8);
        __syncthreads();                     // this workgroup's partial sums are out (bnb_put) before its arrival is counted
        if (threadIdx.x == 0) {
            const int before = atomicAdd(&arrive[a], 1);
            s_last = before == nsplit - 1;
            if (s_last) arrive[a] = 0;       // (nobody else touches this count before the next launch)
        }
        __syncthreads();
        if (s_last) {                        // the same in all threads
            search_step_one(a, states, heap_cube, heap_lb, partials, nsplit, active_in, active_out, cnt_out, parents_out, sse_thresh, h_cube, h_lb, s_bound, &s_n);
        }
    }
}

__global__ void k_search_finish(const SearchState* __restrict__ st, int n, icpx_bnb_found* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SearchState s = st[i];
    icpx_bnb_found f;
    f.opt_error = s.opt;
    f.have_best = s.have_best;
    for (int k = 0; k < 4; k++) f.best[k] = s.best[k];
    f.cubes = s.pops;
    f.status = s.status ? s.status : (s.done ? 0 : 2);      // 2: still running when the steps ran out
    out[i] = f;
}

extern "C" int icpx_bnb_search_run(icpx_ctx* ctx, const icpx_bnb_search* searches, size_t n, float sse_thresh, icpx_bnb_found* found, long long* steps_out) {
    if (!ctx) return ICPX_EINVAL;
    if (steps_out) *steps_out = 0;
    if (n == 0) return ICPX_OK;
    if (!searches || !found) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_search_run: NULL argument");
    if (n > ICPX_SEARCH_MAX) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_bnb_search_run: more than %d searches in one call", ICPX_SEARCH_MAX);
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_search_run: no EDT has been built");
    if (ctx->nd == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_bnb_search_run: no data cloud (call icpx_goicp_set_data first)");
    for (size_t i = 0; i < n; i++)
        if (searches[i].level >= ICPX_MAXROTLEVEL) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_bnb_search_run: search %zu has level %d >= %d", i, searches[i].level, ICPX_MAXROTLEVEL);
    ICPX_TRY(icpx_bind(ctx));
    if (ctx->bnb_bricks) ICPX_TRY(bnb_bricks_ensure(ctx));
    const size_t nd = ctx->nd;
    const int nsplit = bnb_slices(nd);
    DevBuf* sb = ctx->search_buf;
    ICPX_TRY(icpx_reserve(ctx, sb[0], n * sizeof(SearchState)));
    ICPX_TRY(icpx_reserve(ctx, sb[1], n * ICPX_SEARCH_HEAP_CAP * sizeof(float4)));
    ICPX_TRY(icpx_reserve(ctx, sb[2], n * ICPX_SEARCH_HEAP_CAP * sizeof(float)));
    ICPX_TRY(icpx_reserve(ctx, sb[3], 2 * n * sizeof(int)));
    ICPX_TRY(icpx_reserve(ctx, sb[4], 2 * n * sizeof(icpx_bnb_item)));
    ICPX_TRY(icpx_reserve(ctx, sb[5], n * 8 * (size_t)nsplit * 2 * sizeof(double)));
    ICPX_TRY(icpx_reserve(ctx, sb[6], (ICPX_SEARCH_MAX_STEPS + 1 + n) * sizeof(int)));       // list lengths per step, then the arrival counts
    ICPX_TRY(icpx_reserve(ctx, sb[7], n * std::max(sizeof(icpx_bnb_search), sizeof(icpx_bnb_found))));
    SearchState* st = sb[0].as<SearchState>();
    int* active = sb[3].as<int>();
    icpx_bnb_item* parents = sb[4].as<icpx_bnb_item>();
    int* cnt = sb[6].as<int>();
    const size_t io_bytes = n * std::max(sizeof(icpx_bnb_search), sizeof(icpx_bnb_found));
    ICPX_TRY(bnb_pin_reserve(ctx, io_bytes + 64));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));       // an earlier transfer may still use the staging area
    int* pin_cnt = reinterpret_cast<int*>(static_cast<char*>(ctx->bnb_pin) + io_bytes);
    std::memcpy(ctx->bnb_pin, searches, n * sizeof(icpx_bnb_search));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(sb[7].p, ctx->bnb_pin, n * sizeof(icpx_bnb_search), hipMemcpyHostToDevice, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipMemsetAsync(cnt, 0, (ICPX_SEARCH_MAX_STEPS + 1 + n) * sizeof(int), ctx->stream));
    int* arrive = cnt + ICPX_SEARCH_MAX_STEPS + 1;
    hipLaunchKernelGGL(k_search_start, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sb[7].as<icpx_bnb_search>(), (int)n, st, active, parents, cnt);
    auto eval = !ctx->edt_bricks_ready ? k_search_eval_step<0, 1> : ctx->edt_fbricks_ready ? k_search_eval_step<2, 1> : k_search_eval_step<1, 1>;
    // (with few searches left a step lasts as long as one thread's chain of points: several points in flight per thread then)
    auto eval_few = ctx->edt_fbricks_ready ? k_search_eval_step<2, 4> : eval;
    const EdtView view = edt_view(ctx);
    size_t running = n;            // an upper bound on the searches still running
    int step = 0;
    // Steps are enqueued a few at a time; after each lot the length of the list comes back. Few steps per lot at first
    // (most searches end within a handful and the grids shrink with the list), more once only the long searches are left.
    int lot = ctx->search_lot;
    const int max_steps = std::min(ctx->search_max_steps, ICPX_SEARCH_MAX_STEPS);
    while (running > 0 && step < max_steps) {
        const int upto = std::min(step + lot, max_steps);
        const unsigned gx = (unsigned)std::min<size_t>(running, (size_t)ctx->num_cus * 4);
        for (; step < upto; step++) {
            const int in = step & 1, out = in ^ 1;
            hipLaunchKernelGGL(running <= (size_t)ctx->search_few ? eval_few : eval, dim3(gx, (unsigned)nsplit), dim3(256), 0, ctx->stream, view, ctx->data_pts.as<float4>(), nd, ctx->maxrot.as<float>(),
                               parents + (size_t)in * n, cnt + step, nsplit, sb[5].as<double>(), arrive, st, sb[1].as<float4>(), sb[2].as<float>(),
                               active + (size_t)in * n, active + (size_t)out * n, cnt + step + 1, parents + (size_t)out * n, sse_thresh);
        }
        ICPX_TRY(icpx_launch_check(ctx, "k_search_eval_step"));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(pin_cnt, cnt + step, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        running = (size_t)*pin_cnt;
        lot = std::min(2 * lot, ctx->search_lot_max);
    }
    hipLaunchKernelGGL(k_search_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, st, (int)n, sb[7].as<icpx_bnb_found>());
    ICPX_TRY(icpx_launch_check(ctx, "k_search_finish"));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->bnb_pin, sb[7].p, n * sizeof(icpx_bnb_found), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(found, ctx->bnb_pin, n * sizeof(icpx_bnb_found));
    if (steps_out) *steps_out = step;
    return ICPX_OK;
}

extern "C" int icpx_goicp_residuals(icpx_ctx* ctx, const double T[16], float* dist_out) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_goicp_residuals: no EDT has been built");
    if (ctx->nd == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_goicp_residuals: no data cloud (call icpx_goicp_set_data first)");
    if (!T || !dist_out) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_goicp_residuals: NULL argument");
    ICPX_TRY(icpx_bind(ctx));
    const size_t n = ctx->nd;
    ICPX_TRY(icpx_reserve(ctx, ctx->bnb_mindis, n * sizeof(float)));
    Mat34 M;
    for (int k = 0; k < 12; k++) M.m[k] = T[k];
    hipLaunchKernelGGL(k_goicp_residuals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, edt_view(ctx),
                       ctx->data_pts.as<float4>(), n, M, ctx->bnb_mindis.as<float>());
    ICPX_TRY(icpx_launch_check(ctx, "k_goicp_residuals"));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(dist_out, ctx->bnb_mindis.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}
