// Uniform-grid nearest-neighbour search and the fused ICP iteration kernel.
//
// Stands in for the reference's kd-tree (src/kdtree.cpp:58-150) and for the per-iteration loops of
// SingleThreadIcp (kdtree_search src/SingleThreadIcp.cpp:186-214, cal_covarance :216-247, the residual
// of best_fit_transform :281-287). Results are exact: every candidate distance is the reference's
// double-precision expression ((dx*dx + dy*dy) + dz*dz) without fused multiply-add (this file is built
// with -ffp-contract=off), ties resolve to the lowest target index, and cells / subtrees are skipped only
// when a conservative lower bound on their distance is strictly larger than the best distance so far.
//
// Layout: target points sorted by cell (stable, so ascending original index inside a cell) as
// float4 {x, y, z, bits(index)}; cell_start[cell] = position of the cell's first point in that order
// (cell_start[cell + 1] its end; cells are numbered (x * dy + y) * dz + z, so the points of a run of cells along z
// are one contiguous run). Queries of the ICP step are the source cloud in Morton order, so the 64 queries of a
// wave lie in a few neighbouring cells.
#include "icpx_ctx.h"
#include "icpx_solve.h"

#include <rocprim/device/device_scan.hpp>

#include <cfloat>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <vector>

int icpx_sort_pairs(icpx_ctx* ctx, size_t n, unsigned end_bit);
int icpx_sort_pairs64(icpx_ctx* ctx, size_t n, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned* vals_in,
                      unsigned* vals_out);
int icpx_sort_reserve(icpx_ctx* ctx, size_t n);

namespace {

constexpr int MAXLEV = 11;

// Level 0 is the cell grid itself; level l >= 1 halves every dimension (rounding up), up to a single root
// node, and keeps one byte per node: bit (di << 2 | dj << 1 | dk) is set when child (2i + di, 2j + dj,
// 2k + dk) of the level below holds a point. The search uses the pyramid to skip empty space.
struct GridView {
    int dx, dy, dz;
    float minx, miny, minz, h, inv;
    const int* __restrict__ start;             // ncells + 1 positions into pts
    const float4* __restrict__ pts;
    const unsigned char* __restrict__ masks;   // child masks of all levels >= 1, level l at masks + loff[l]
    const unsigned* __restrict__ loff;          // MAXLEV offsets into masks (device memory)
    int nlev;                                   // number of levels including level 0
    // candidate table: for every block of 2^cshift cells per axis, the position (in pts) of a target point near the
    // block's centre (-1 only for an empty target); see k_cand_seed
    const int* __restrict__ cand;
    int cdx, cdy, cdz, cshift;
};

// Cells along one axis at level l: ceil(d / 2^l).
__device__ __forceinline__ int lev_dim(int d, int l) { return (d + (1 << l) - 1) >> l; }

__device__ __forceinline__ int cell_coord(float v, float mn, float inv, int dim) {
    const float f = floorf((v - mn) * inv);
    return (int)fminf(fmaxf(f, 0.f), (float)(dim - 1));
}

__global__ void k_cell_keys(const float4* __restrict__ pts, size_t n, GridView g, unsigned* __restrict__ keys, unsigned* __restrict__ vals) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int cx = cell_coord(p.x, g.minx, g.inv, g.dx), cy = cell_coord(p.y, g.miny, g.inv, g.dy), cz = cell_coord(p.z, g.minz, g.inv, g.dz);
    keys[i] = (unsigned)((cx * g.dy + cy) * g.dz + cz);
    vals[i] = (unsigned)i;
}

// Points per cell from the sorted keys (one add per run of equal keys) and the number of occupied cells; the
// exclusive prefix sum of the counts is cell_start.
__global__ void k_cell_counts(const unsigned* __restrict__ keys, size_t n, int* __restrict__ count, unsigned* __restrict__ distinct) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = keys[i];
    if (i != 0 && keys[i - 1] == k) return;
    size_t e = i + 1;
    while (e < n && keys[e] == k) e++;
    count[k] = (int)(e - i);
    atomicAdd(distinct, 1u);
}

__global__ void k_gather_pts(const float4* __restrict__ src, const unsigned* __restrict__ order, size_t n, float4* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[order[i]];
}

// Sets the bit of child (cx, cy, cz) in the mask byte of its parent; `up` is the 4-byte aligned first word of
// the parent level (bytes have no atomics: the OR goes to the word that holds the byte).
__device__ __forceinline__ void pyr_mark(unsigned* __restrict__ up, int cx, int cy, int cz, int uy, int uz) {
    const size_t parent = ((size_t)(cx >> 1) * uy + (cy >> 1)) * uz + (cz >> 1);
    const unsigned bit = (unsigned)(((cx & 1) << 2) | ((cy & 1) << 1) | (cz & 1));
    atomicOr(up + (parent >> 2), (1u << bit) << (8 * (parent & 3)));
}

__global__ void k_pyr_from_cells(const int* __restrict__ start, int dx, int dy, int dz, unsigned* __restrict__ up, int uy, int uz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)dx * dy * dz) return;
    if (start[i + 1] <= start[i]) return;
    pyr_mark(up, (int)(i / ((size_t)dz * dy)), (int)((i / dz) % dy), (int)(i % dz), uy, uz);
}

__global__ void k_pyr_from_masks(const unsigned char* __restrict__ lo, int dx, int dy, int dz, unsigned* __restrict__ up, int uy, int uz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)dx * dy * dz) return;
    if (!lo[i]) return;
    pyr_mark(up, (int)(i / ((size_t)dz * dy)), (int)((i / dz) % dy), (int)(i % dz), uy, uz);
}

// Search results. Candidates are ordered by (squared distance, index); bound() is the distance beyond which
// nothing can enter any more.
struct Best {               // the single nearest; pos = its position in the cell-sorted order (-1: none)
    double d2;
    int idx, pos;
    __device__ __forceinline__ void init() { d2 = INFINITY; idx = 0x7fffffff; pos = -1; }
    __device__ __forceinline__ double bound() const { return d2; }
    __device__ __forceinline__ void offer(double d, int id, int at) {
        if (d < d2 || (d == d2 && id < idx)) { d2 = d; idx = id; pos = at; }
    }
};

template <int K>
struct BestK {              // the K nearest, ascending; unused slots hold (INFINITY, 0x7fffffff); a point is offered at most once
    double d2[K];
    int idx[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < K; k++) { d2[k] = INFINITY; idx[k] = 0x7fffffff; }
    }
    __device__ __forceinline__ double bound() const { return d2[K - 1]; }
    __device__ __forceinline__ void offer(double d, int id, int = 0) {
        if (!(d < d2[K - 1] || (d == d2[K - 1] && id < idx[K - 1]))) return;
        // one pass of insertion from the back: shift the larger entries up, drop the candidate in its place
#pragma unroll
        for (int k = K - 1; k >= 0; k--) {
            const bool before_prev = k > 0 && (d < d2[k - 1] || (d == d2[k - 1] && id < idx[k - 1]));
            if (before_prev) { d2[k] = d2[k - 1]; idx[k] = idx[k - 1]; }
            else { d2[k] = d; idx[k] = id; break; }
        }
    }
};

// The query as the search sees it in single precision. With qf = (float)q and u = 2^-24 the float sum
// s = dx^2 + dy^2 + dz^2 of float differences to a target point t satisfies |q - t| >= sqrt(s) (1 - 3.1 u) - e,
// e = u (|qx| + |qy| + |qz|) (one rounding of q, one of each difference, three of the sum), so a point with
// s > (sqrt(B) + e)^2 (1 + 2^-20) is strictly farther than squared distance B. 2 sqrt(B) <= B / c + c for any c > 0
// (c: the cell size), which turns the right-hand side into B k1 + k2 with per-query constants and no square root.
struct QueryF {
    float x, y, z;
    float k1, k2;       // a point with float squared distance above fma(B, k1, k2) cannot be as near as B
    float margin;       // shrinks the boxes: the slop of the float cell assignment plus that of the float box bound
    __device__ __forceinline__ QueryF(double qx, double qy, double qz, double h) {
        x = (float)qx; y = (float)qy; z = (float)qz;
        const double sum = fabs(qx) + fabs(qy) + fabs(qz), e = 0x1p-24 * sum;
        k1 = (float)((1.0 + e / h) * (1.0 + 0x1p-19));
        k2 = (float)((e * h + e * e) * (1.0 + 0x1p-19));
        margin = (float)((1e-3 * h + 1.2e-6 * sum) * (1.0 + 0x1p-20));
    }
    __device__ __forceinline__ float point_threshold(double best) const { return fmaf((float)best, k1, k2); }
};

// What a float lower bound has to exceed to prove its box strictly farther than `best`: the bound below comes out at
// most a factor 1 + 2^-20 above the exact one, the conversion at most 2^-24 below `best`.
__device__ __forceinline__ float box_threshold(double best) { return (float)best * (1.f + 0x1p-19f); }

template <class R>
__device__ __forceinline__ void scan_cell(const GridView& g, int cell, double qx, double qy, double qz, const QueryF&, R& b) {
    const int first = g.start[cell], last = g.start[cell + 1];
    for (int k = first; k < last; k++) {
        const float4 p = g.pts[k];
        const double dx = qx - (double)p.x, dy = qy - (double)p.y, dz = qz - (double)p.z;
        const double d = (dx * dx + dy * dy) + dz * dz;
        b.offer(d, __float_as_int(p.w), k);
    }
}

// Single nearest: the points are compared in float first, the exact distance is formed only for those that the float
// value cannot rule out.
__device__ __forceinline__ void scan_cell(const GridView& g, int cell, double qx, double qy, double qz, const QueryF& f, Best& b) {
    const int first = g.start[cell], last = g.start[cell + 1];
    float thr = f.point_threshold(b.d2);
    // four points per round, their loads issued together (a cell holds about five points: one wait for memory per
    // round instead of one per point; positions past the end repeat the last point and are not offered)
    for (int k0 = first; k0 < last; k0 += 4) {
        float4 p4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p4[u] = g.pts[min(k0 + u, last - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float4 p = p4[u];
            const float fx = f.x - p.x, fy = f.y - p.y, fz = f.z - p.z;
            const float sf = (fx * fx + fy * fy) + fz * fz;
            if (k0 + u >= last || sf > thr) continue;
            const double dx = qx - (double)p.x, dy = qy - (double)p.y, dz = qz - (double)p.z;
            const double d = (dx * dx + dy * dy) + dz * dz;
            b.offer(d, __float_as_int(p.w), k0 + u);
            thr = f.point_threshold(b.d2);
        }
    }
}

// Distance from coordinate q to the interval [o + i side, o + (i + 1) side], shrunk by `margin`, squared; 0 inside.
// In float: the ends carry one rounding each (2^-24 of their size), q one, the difference one; together less than
// 2^-23 (|q| + gap), which the margin (absolute part) and box_threshold (relative part) allow for.
__device__ __forceinline__ float axis_gap2(float q, float o, float side, int i, float margin) {
    const float lo = fmaf((float)i, side, o), hi = fmaf((float)(i + 1), side, o);
    const float g = fmaxf(fmaxf(lo - q, q - hi) - margin, 0.f);
    return g * g;
}

// Squared distance from q to the box of cell (i,j,k) at a level whose cells have side `side`, shrunk by
// the margin so that (against box_threshold) it is a valid lower bound for every point filed under that cell.
__device__ __forceinline__ float box_lb2(const QueryF& f, float ox, float oy, float oz, float side, int i, int j, int k) {
    return (axis_gap2(f.x, ox, side, i, f.margin) + axis_gap2(f.y, oy, side, j, f.margin)) + axis_gap2(f.z, oz, side, k, f.margin);
}

// Children of a node in visiting order for a query whose nearest octant is `near`: bit t of the result is
// child near ^ j(t), with j = 0, 1, 2, 4, 3, 5, 6, 7 (by number of axes in which the child lies on the far
// side). XOR of the child number is a bit permutation of the mask: swap neighbours, pairs, halves.
__device__ __forceinline__ unsigned pyr_visit_order(unsigned m, int near) {
    if (near & 1) m = ((m & 0x55u) << 1) | ((m >> 1) & 0x55u);
    if (near & 2) m = ((m & 0x33u) << 2) | ((m >> 2) & 0x33u);
    if (near & 4) m = ((m & 0x0fu) << 4) | ((m >> 4) & 0x0fu);
    const unsigned x = ((m >> 3) ^ (m >> 4)) & 1u;
    return m ^ (x << 3) ^ (x << 4);
}

// ---- the search: eight lanes per query --------------------------------------------------------------------
// Exact nearest neighbour(s). First the query's own cell and the 26 around it; that is final when the bound
// (the distance of the nearest, or k-th nearest, found) is smaller than the distance to everything outside that
// 3x3x3 block. Otherwise (query far from the cloud, or a large empty gap) a walk down the occupancy pyramid,
// nearer children first, skipping empty subtrees (one mask byte per node tells which children hold points) and
// those whose box is farther than the bound.
// One lane per query would leave a long chain of dependent loads (mask, cell range, points, next mask ...) and,
// for clouds of some ten thousand points, fewer waves than the GPU has SIMDs. So the eight lanes of an
// aligned group ("octet") share one query: the 27 cells of the first block are dealt out over the lanes,
// and in the walk lane c owns child c of the node being expanded, so the eight bounds are computed, and at
// the lowest level the up to eight cells scanned, side by side. All lanes of an octet hold the same walk
// state; the result (minimum over the lanes by (distance, index)) is the same as the single-lane search's.

// Minimum by (d2, idx) over the octet; every lane gets it.
__device__ __forceinline__ Best octet_min(Best b) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const double d = __shfl_xor(b.d2, o);
        const int id = __shfl_xor(b.idx, o), at = __shfl_xor(b.pos, o);
        if (d < b.d2 || (d == b.d2 && id < b.idx)) { b.d2 = d; b.idx = id; b.pos = at; }
    }
    return b;
}

__device__ __forceinline__ void octet_merge(Best& b) { b = octet_min(b); }

// k nearest for an octet: `g` is the octet's k best so far (the same in all eight lanes, it gives the bound),
// `l` what this lane has found since the last merge. A point is met by one lane only (every cell is scanned once),
// so the lists are disjoint and the merge is a k-round selection: every round takes the smaller of g's head and
// the best head among the lanes' lists (octet_min) and removes it from where it came.
template <int K>
struct OctK {
    BestK<K> g, l;
    __device__ __forceinline__ void init() { g.init(); l.init(); }
    __device__ __forceinline__ double bound() const { return g.d2[K - 1]; }
    __device__ __forceinline__ void offer(double d, int id, int = 0) {
        if (!(d < g.d2[K - 1] || (d == g.d2[K - 1] && id < g.idx[K - 1]))) return;
        l.offer(d, id);
    }
};

template <int K>
__device__ __forceinline__ void octet_merge(OctK<K>& b) {
    // nothing new in any lane: g stands
    if (((__ballot(b.l.idx[0] != 0x7fffffff) >> (threadIdx.x & 56)) & 0xffull) == 0) return;
    BestK<K> out;
#pragma unroll
    for (int r = 0; r < K; r++) {
        Best head;
        head.d2 = b.l.d2[0]; head.idx = b.l.idx[0]; head.pos = -1;
        const Best m = octet_min(head);
        const bool from_g = b.g.d2[0] < m.d2 || (b.g.d2[0] == m.d2 && b.g.idx[0] <= m.idx);
        out.d2[r] = from_g ? b.g.d2[0] : m.d2;
        out.idx[r] = from_g ? b.g.idx[0] : m.idx;
        const bool mine = !from_g && b.l.idx[0] == m.idx;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool last = k == K - 1;
            const double gd = last ? (double)INFINITY : b.g.d2[last ? k : k + 1], ld = last ? (double)INFINITY : b.l.d2[last ? k : k + 1];
            const int gi = last ? 0x7fffffff : b.g.idx[last ? k : k + 1], li = last ? 0x7fffffff : b.l.idx[last ? k : k + 1];
            b.g.d2[k] = from_g ? gd : b.g.d2[k]; b.g.idx[k] = from_g ? gi : b.g.idx[k];
            b.l.d2[k] = mine ? ld : b.l.d2[k]; b.l.idx[k] = mine ? li : b.l.idx[k];
        }
    }
    b.g = out;
    b.l.init();
}

constexpr int NN_OCTETS = 32;   // octets per 256-thread block
constexpr int NN_STACK = 72;    // 1 + 7 entries for each of the (at most ten) levels above the cells, rounded up

// `snode`, `slb`: the walk's stack in LDS, NN_STACK entries per octet, entry k of octet `oc` at [k * nocts + oc];
// `b`: the result so far (Best or OctK<K>), initialised, the same in all lanes: empty, or holding candidates
// that are real target points, e.g. the correspondence of the previous ICP iteration, whose distance then prunes
// the search from the start; the result does not depend on them (equal distances still go to the lower index
// because cells are only skipped when their bound is strictly larger).
template <class R>
__device__ R nn_search8(const GridView& g, double qx, double qy, double qz, unsigned* snode, float* slb, const int oc, const int nocts,
                        const unsigned* loff, R b) {
    const int c = (int)(threadIdx.x & 7);
    const int cx = cell_coord((float)qx, g.minx, g.inv, g.dx), cy = cell_coord((float)qy, g.miny, g.inv, g.dy),
              cz = cell_coord((float)qz, g.minz, g.inv, g.dz);
    const double h = (double)g.h;
    const double margin = 1e-3 * h + 1e-6 * (fabs(qx) + fabs(qy) + fabs(qz));
    const double ox = (double)g.minx, oy = (double)g.miny, oz = (double)g.minz;
    const QueryF f(qx, qy, qz, h);
    float bt = box_threshold(b.bound());
    // cells c, c + 8, c + 16, c + 24 of the 3x3x3 block
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int cc = c + 8 * r;
        const int i = cx + cc / 9 - 1, j = cy + (cc / 3) % 3 - 1, k = cz + cc % 3 - 1;
        if (cc < 27 && i >= 0 && i < g.dx && j >= 0 && j < g.dy && k >= 0 && k < g.dz &&
            !(box_lb2(f, g.minx, g.miny, g.minz, g.h, i, j, k) > bt)) {
            scan_cell(g, (i * g.dy + j) * g.dz + k, qx, qy, qz, f, b);
            bt = box_threshold(b.bound());
        }
    }
    octet_merge(b);
    double outside = INFINITY;
    if (cx + 2 <= g.dx - 1) { const double t = fmax(ox + (cx + 2) * h - qx - margin, 0.0); outside = fmin(outside, t * t); }
    if (cx - 2 >= 0)        { const double t = fmax(qx - (ox + (cx - 1) * h) - margin, 0.0); outside = fmin(outside, t * t); }
    if (cy + 2 <= g.dy - 1) { const double t = fmax(oy + (cy + 2) * h - qy - margin, 0.0); outside = fmin(outside, t * t); }
    if (cy - 2 >= 0)        { const double t = fmax(qy - (oy + (cy - 1) * h) - margin, 0.0); outside = fmin(outside, t * t); }
    if (cz + 2 <= g.dz - 1) { const double t = fmax(oz + (cz + 2) * h - qz - margin, 0.0); outside = fmin(outside, t * t); }
    if (cz - 2 >= 0)        { const double t = fmax(qz - (oz + (cz - 1) * h) - margin, 0.0); outside = fmin(outside, t * t); }
    if (outside > b.bound()) return b;
    const int top = g.nlev - 1;
    if (top == 0) return b;

    // Walk: a stack of nodes still to expand, each with the bound of its box; the node on top is the nearest
    // of those pushed last. Expanding a node: lane c bounds child c; at level 1 the children are cells and
    // every lane scans its own, above that the children that can still hold a nearer point are pushed, the
    // farthest first (in the order of visit_order: the octant of the query, then one axis flipped, ...).
    // A node whose bound has been overtaken by the best distance since it was pushed is dropped when popped.
    // Depth: one pop and at most eight pushes per level, NN_STACK covers the ten levels above the cells.
    int sp = 1;
    snode[oc] = (unsigned)top << 27;
    slb[oc] = 0.f;
    bt = box_threshold(b.bound());
    while (sp > 0) {
        sp--;
        const unsigned node = snode[sp * nocts + oc];
        if (slb[sp * nocts + oc] > bt) continue;
        const int l = (int)(node >> 27), ni = (int)((node >> 18) & 511), nj = (int)((node >> 9) & 511), nk = (int)(node & 511);
        const unsigned m = g.masks[loff[l] + ((size_t)ni * lev_dim(g.dy, l) + nj) * lev_dim(g.dz, l) + nk];
        const int cl = l - 1;
        const float cside = g.h * (float)(1 << cl);   // exact: a power of two times h
        const int ci = 2 * ni + (c >> 2), cj = 2 * nj + ((c >> 1) & 1), ck = 2 * nk + (c & 1);
        const bool has = (m >> c) & 1u;
        const float lb = has ? box_lb2(f, g.minx, g.miny, g.minz, cside, ci, cj, ck) : INFINITY;
        const bool ok = has && !(lb > bt);
        if (cl == 0) {
            // the children are cells: every lane scans its own (those of the first block are done already)
            const bool seen = ci >= cx - 1 && ci <= cx + 1 && cj >= cy - 1 && cj <= cy + 1 && ck >= cz - 1 && ck <= cz + 1;
            if (ok && !seen) scan_cell(g, (ci * g.dy + cj) * g.dz + ck, qx, qy, qz, f, b);
            octet_merge(b);
            bt = box_threshold(b.bound());
            continue;
        }
        const unsigned cand = (unsigned)(__ballot(ok) >> (threadIdx.x & 56)) & 0xffu;
        if (cand == 0) continue;
        const int near = (f.x >= fmaf((float)(2 * ni + 1), cside, g.minx) ? 4 : 0) | (f.y >= fmaf((float)(2 * nj + 1), cside, g.miny) ? 2 : 0) |
                         (f.z >= fmaf((float)(2 * nk + 1), cside, g.minz) ? 1 : 0);
        const unsigned vis = pyr_visit_order(cand, near);
        const int jc = c ^ near, t = jc == 3 ? 4 : jc == 4 ? 3 : jc;          // this lane's place in the visiting order
        const int n = __popc(cand);
        if (ok) {
            const int at = sp + n - 1 - __popc(vis & ((1u << t) - 1u));
            snode[at * nocts + oc] = ((unsigned)cl << 27) | ((unsigned)ci << 18) | ((unsigned)cj << 9) | (unsigned)ck;
            slb[at * nocts + oc] = lb;
        }
        sp += n;
    }
    return b;
}

__device__ __forceinline__ Best best_empty() { Best b; b.init(); return b; }

// The previous correspondence of a source point as the first candidate: `seed` holds, per source point, the position
// of that target point in the cell-sorted order (-1: none). Gives (distance to it now, its index, that position) and
// the point itself; empty when there is none or the position does not belong to the current target.
__device__ __forceinline__ Best best_seed(const GridView& g, const int* __restrict__ seed, size_t i, size_t nt, double qx, double qy, double qz,
                                          float4* point = nullptr) {
    Best b;
    b.init();
    if (seed) {
        const int at = seed[i];
        if (at >= 0 && (size_t)at < nt) {
            const float4 t = g.pts[at];
            const double dx = qx - (double)t.x, dy = qy - (double)t.y, dz = qz - (double)t.z;
            b.d2 = (dx * dx + dy * dy) + dz * dz;
            b.idx = __float_as_int(t.w);
            b.pos = at;
            if (point) *point = t;
        }
    }
    return b;
}

// The same for the kernels of an ICP pass, which keep the pair as a copy of the target point itself (x, y, z, index bits;
// index bits < 0: none) by source point: one coalesced read where the position costs a read and then a gather that the
// whole tile waits for. Such a candidate has no position: Best::pos is PAIR_POS. `point` receives the copy.
constexpr int PAIR_POS = 0x7ffffffe;
__device__ __forceinline__ Best best_pair(const float4* __restrict__ pair, size_t i, double qx, double qy, double qz, float4& point) {
    Best b;
    b.init();
    const float4 t = pair[i];
    point = t;
    if (__float_as_int(t.w) >= 0) {
        const double dx = qx - (double)t.x, dy = qy - (double)t.y, dz = qz - (double)t.z;
        b.d2 = (dx * dx + dy * dy) + dz * dz;
        b.idx = __float_as_int(t.w);
        b.pos = PAIR_POS;
    }
    return b;
}
// Best::offer that also keeps the point of the best
__device__ __forceinline__ void offer_point(Best& b, float4& point, double d, const float4& t, int at) {
    if (d < b.d2 || (d == b.d2 && __float_as_int(t.w) < b.idx)) { b.d2 = d; b.idx = __float_as_int(t.w); b.pos = at; point = t; }
}

__device__ __forceinline__ void cand_lookup(const GridView& g, double qx, double qy, double qz, Best& b, float4* point = nullptr);   // the candidate table, below

// K nearest target points of every query, ascending by (squared distance, index); slots beyond the number
// of target points get index -1 and an infinite distance.
template <int K>
__global__ __launch_bounds__(256) void k_knn(GridView g, const double* __restrict__ q, size_t n, int kout, int* __restrict__ idx,
                                             double* __restrict__ d2) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ float slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;   // eight lanes per query
    if (i >= n) return;
    OctK<K> b0;
    b0.init();
    const OctK<K> b = nn_search8(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], snode, slb, (int)(threadIdx.x >> 3), NN_OCTETS, s_loff, b0);
    if ((threadIdx.x & 7) != 0) return;
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (k < kout) {
            if (idx) idx[i * kout + k] = b.g.idx[k] == 0x7fffffff ? -1 : b.g.idx[k];
            if (d2) d2[i * kout + k] = b.g.d2[k];
        }
    }
}

__global__ __launch_bounds__(256) void k_nn(GridView g, const double* __restrict__ q, size_t n, int* __restrict__ idx, double* __restrict__ d2) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ float slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;   // eight lanes per query
    if (i >= n) return;
    Best b0 = best_empty();
    cand_lookup(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], b0);      // a real point to start from: its distance prunes the walk
    const Best b = nn_search8(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], snode, slb, (int)(threadIdx.x >> 3), NN_OCTETS, s_loff, b0);
    if ((threadIdx.x & 7) == 0) {
        if (idx) idx[i] = b.idx;
        if (d2) d2[i] = b.d2;
    }
}

__global__ __launch_bounds__(256) void k_nn_distance(GridView g, const float* __restrict__ q, size_t n, float* __restrict__ out) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ float slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (i >= n) return;
    Best b0 = best_empty();
    cand_lookup(g, (double)q[3 * i], (double)q[3 * i + 1], (double)q[3 * i + 2], b0);
    const Best b = nn_search8(g, (double)q[3 * i], (double)q[3 * i + 1], (double)q[3 * i + 2], snode, slb, (int)(threadIdx.x >> 3), NN_OCTETS, s_loff, b0);
    if ((threadIdx.x & 7) == 0) out[i] = (float)__dsqrt_rn(b.d2);
}

// ---- candidate table -------------------------------------------------------------------------------------------
// A query that comes without a previous pair (the first pass of an ICP run, every plain nearest-neighbour query) needs a
// real target point to start from: its distance bounds the search. A greedy descent of the pyramid finds one at the cost
// of a chain of dependent reads per level and may land far from the nearest point. The table holds, for every block of
// 2^cshift cells per axis over the grid's box, a target point near the block's centre: the blocks that hold points are
// seeded with one of theirs, then the entries spread by jump flooding (each block looks at the entries of the 26 blocks
// `step` away and keeps the one whose point is nearest to its own centre; steps halve down to 1, and 1 once more). What
// it gives is a candidate about as far as the nearest point plus the block's size; exactness is the search's business.
__global__ void k_cand_seed(GridView g, int* __restrict__ tab) {
    const size_t n = (size_t)g.cdx * g.cdy * g.cdz, e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int bk = (int)(e % g.cdz), bj = (int)((e / g.cdz) % g.cdy), bi = (int)(e / ((size_t)g.cdz * g.cdy));
    const int side = 1 << g.cshift;
    int found = -1;
    for (int i = bi * side; i < min((bi + 1) * side, g.dx) && found < 0; i++)
        for (int j = bj * side; j < min((bj + 1) * side, g.dy) && found < 0; j++) {
            const size_t row = ((size_t)i * g.dy + j) * g.dz;
            const int k0 = bk * side, k1 = min((bk + 1) * side, g.dz);
            const int a = g.start[row + k0], b = g.start[row + k1];
            if (b > a) found = a;
        }
    tab[e] = found;
}

__global__ void k_cand_flood(GridView g, const int* __restrict__ in, int* __restrict__ out, int step) {
    const size_t n = (size_t)g.cdx * g.cdy * g.cdz, e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int bk = (int)(e % g.cdz), bj = (int)((e / g.cdz) % g.cdy), bi = (int)(e / ((size_t)g.cdz * g.cdy));
    const float side = g.h * (float)(1 << g.cshift);
    const float cx = fmaf((float)bi + 0.5f, side, g.minx), cy = fmaf((float)bj + 0.5f, side, g.miny), cz = fmaf((float)bk + 0.5f, side, g.minz);
    int best = in[e];
    float bd = INFINITY;
    if (best >= 0) { const float4 t = g.pts[best]; const float dx = t.x - cx, dy = t.y - cy, dz = t.z - cz; bd = (dx * dx + dy * dy) + dz * dz; }
    for (int di = -1; di <= 1; di++)
        for (int dj = -1; dj <= 1; dj++)
            for (int dk = -1; dk <= 1; dk++) {
                const int i = bi + di * step, j = bj + dj * step, k = bk + dk * step;
                if ((di | dj | dk) == 0 || i < 0 || j < 0 || k < 0 || i >= g.cdx || j >= g.cdy || k >= g.cdz) continue;
                const int c = in[((size_t)i * g.cdy + j) * g.cdz + k];
                if (c < 0 || c == best) continue;
                const float4 t = g.pts[c];
                const float dx = t.x - cx, dy = t.y - cy, dz = t.z - cz, d = (dx * dx + dy * dy) + dz * dz;
                if (d < bd) { bd = d; best = c; }
            }
    out[e] = best;
}

// The table's candidate for q, offered to b with its exact distance.
__device__ __forceinline__ void cand_lookup(const GridView& g, double qx, double qy, double qz, Best& b, float4* point) {
    if (!g.cand) return;
    const int i = cell_coord((float)qx, g.minx, g.inv, g.dx) >> g.cshift, j = cell_coord((float)qy, g.miny, g.inv, g.dy) >> g.cshift,
              k = cell_coord((float)qz, g.minz, g.inv, g.dz) >> g.cshift;
    const int at = g.cand[((size_t)i * g.cdy + j) * g.cdz + k];
    if (at < 0) return;
    const float4 t = g.pts[at];
    const double dx = qx - (double)t.x, dy = qy - (double)t.y, dz = qz - (double)t.z;
    const double d = (dx * dx + dy * dy) + dz * dz;
    if (point) offer_point(b, *point, d, t, at);
    else b.offer(d, __float_as_int(t.w), at);
}

struct Mat34 { double m[12]; };

// ---- one ICP pass, a wave per tile of 64 source points ------------------------------------------------------
// p = T * [src;1] (multiply then add, k = 0..3 from a zero accumulator) and its nearest target point q, left in pair[]
// as a position in the sorted target points; k_icp_sums then forms the sums {p, q, p q^T, |p-q|^2}.
//
// The source cloud is in Morton order, so the 64 points of a wave fall into a few neighbouring cells. Every lane names
// the block of cells it needs (see the kernel: the cells its previous pair bounds, or its own cell and the 26 around it);
// the wave takes the box that holds all blocks and copies the target points filed under it into LDS (the box is a set of
// cell columns along z; of each column the run of cells some lane asked for is one contiguous run of the sorted points:
// two reads of cell_start and one run of 16-byte loads per column; TILE_CAP points at a time). Every lane then compares
// its query with all staged points, read from LDS as broadcasts, in float: with qf = (float)q the float sum
// s = dx^2 + dy^2 + dz^2 of float differences satisfies |q - t| >= sqrt(s) (1 - 3.1 u) - u |q|, u = 2^-24 (one
// rounding of q, one of each difference, three of the sum), so a candidate with s above
// thr(d) = (sqrt(d) + u (|qx| + |qy| + |qz|))^2 (1 + 2^-20), rounded up to float, is strictly farther than squared
// distance d and cannot even tie with it. The lane keeps the smallest s, its candidate, and the second smallest s
// (with a few low bits of s given up to carry the candidate's number, which widens the test by 2^-12);
// the candidate's exact double distance d then shows, by thr(d), whether any other candidate could be as near
// (hardly ever one can). What a lane finds is final when no other candidate can and nothing outside its block can be
// nearer. The other lanes (two candidates too close to call in float; queries far from the cloud, as in the first
// passes of a badly aligned start), and whole tiles whose box has more than TILE_COLS columns or TILE_MAX_PTS points,
// are left to k_icp_ball or k_icp_walk, which continue from the candidate they hold.
constexpr int TILE_CAP = 512;            // points staged at a time: 8 KiB of LDS (a power of two: slots are packed into TILE_CAP - 1 bits)
constexpr int TILE_COLS = 256;           // most columns of a box
#ifndef ICPX_TILE_NEAR
#define ICPX_TILE_NEAR 4.0f
#endif
constexpr float TILE_NEAR = ICPX_TILE_NEAR;        // a lane whose previous pair is nearer than this many cells asks only for the cells that pair bounds
                                         // (1 M pair, compact tiles, 30 passes: 20.0 / 18.5 / 17.7 / 17.0 / 17.0 ms for 1.5 / 2 / 3 / 4 / 5 cells here and in BALL_MIN_REACH)
#ifndef ICPX_TILE_MAX_PTS
#define ICPX_TILE_MAX_PTS 1536
#endif
constexpr int TILE_MAX_PTS = ICPX_TILE_MAX_PTS;       // most points of a box: a few times the usual number, so that no tile takes many times as long as the others
constexpr int WALK_SHARDS = 64;          // parts of the walk queue and of the far-tile queue (one wave's lanes read the 64 counts)
constexpr int BALL_MIN_LANES = 8;        // a tile with at least this many open points goes to k_icp_ball, fewer are walked one by one
constexpr int BALL_MAX_COLS = 16384;     // most cell columns under a tile's ball; beyond that the points are left to the walk
constexpr int BALL_MAX_PTS = 4096;       // most points staged for a tile (16 times as many when far tiles abound): a wave with a fuller ball ends up as
                                         // the kernel's tail, and its points are cheaper to walk one by one (1 M pair, 30 passes: 25.3 / 23.4 / 22.5 /
                                         // 22.8 / 23.6 ms with 2 048 / 3 072 / 4 096 / 8 192 / 16 384)
#ifndef ICPX_BALL_MIN_REACH
#define ICPX_BALL_MIN_REACH 4.f
#endif
constexpr float BALL_MIN_REACH = ICPX_BALL_MIN_REACH;    // a query at least this many cells from its candidate counts as far from the surface
#ifndef ICPX_SOLO_MIN
#define ICPX_SOLO_MIN 0.f
#endif
constexpr float SOLO_MIN_REACH = ICPX_SOLO_MIN;  // single open queries at least this many cells from their candidate (1 M pair, 30 passes: 14.0 / 13.6 / 12.8 / 12.6 ms for 3 / 2 / 1 / 0), and ...
#ifndef ICPX_SOLO_MAX
#define ICPX_SOLO_MAX 40.f
#endif
constexpr float SOLO_MAX_REACH = ICPX_SOLO_MAX;  // ... less than this many, get a wave of their own (k_icp_solo) instead of a pyramid walk
constexpr float BALL_MAX_SPREAD = 12.f;  // most cells from the tile's centre to one of its queries, for the same reason

// Minimum / maximum over the lanes of the wave (all of them active), by the device library's cross-lane reduction:
// register-to-register row operations, where six rounds of __shfl_xor go through the LDS crossbar one after the other.
__device__ __forceinline__ int wave_min(int v) { return __ockl_wfred_min_i32(v); }
__device__ __forceinline__ int wave_max(int v) { return __ockl_wfred_max_i32(v); }

__device__ __forceinline__ void icp_transform(const Mat34& T, const float4 s, double p[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) {
        double a = 0.0;
        a += T.m[4 * r + 0] * (double)s.x;
        a += T.m[4 * r + 1] * (double)s.y;
        a += T.m[4 * r + 2] * (double)s.z;
        a += T.m[4 * r + 3];
        p[r] = a;
    }
}

__device__ __forceinline__ void icp_accumulate(double acc[ICPX_NSUMS], const double p[3], const float4 q, double d2) {
    const double q0 = (double)q.x, q1 = (double)q.y, q2 = (double)q.z;
    acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
    acc[3] += q0; acc[4] += q1; acc[5] += q2;
    acc[6] += p[0] * q0; acc[7] += p[0] * q1; acc[8] += p[0] * q2;
    acc[9] += p[1] * q0; acc[10] += p[1] * q1; acc[11] += p[1] * q2;
    acc[12] += p[2] * q0; acc[13] += p[2] * q1; acc[14] += p[2] * q2;
    acc[15] += d2;
}

// A CU admits at most eight workgroups at a time, so single-wave workgroups would leave it at two waves per SIMD:
// the tile kernels run TILE_WAVES independent waves per workgroup (nothing is shared between them and they never
// meet at a workgroup barrier; inside a wave the LDS executes its instructions in order, so wave_sync only has to
// keep the compiler from moving LDS accesses across it).
constexpr int TILE_WAVES = 2;
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// This wave's number among all waves of the launch, and their count.
__device__ __forceinline__ unsigned icp_wave_id() { return blockIdx.x * TILE_WAVES + (threadIdx.x >> 6); }

// Sum of every accumulator over the wave, written by lane 0 as this wave's partial sums.
__device__ __forceinline__ void icp_write_partials(double acc[ICPX_NSUMS], double* __restrict__ partials, int lane) {
#pragma unroll
    for (int k = 0; k < ICPX_NSUMS; k++) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) partials[(size_t)k * gridDim.x * TILE_WAVES + icp_wave_id()] = v;      // component-major: k_reduce_sums reads runs
    }
}

// Tile number tt of a wave. Workgroups b, b + 8, b + 16 ... run on one XCD (round-robin dispatch): the waves of such
// a group share one contiguous run of tiles, dealt out in turn (tile tt of the group's j-th wave is number
// tt * (waves in the group) + j of the run). Waves that are resident together then work on neighbouring tiles, whose
// cell columns overlap, out of one L2, and a stretch of expensive tiles (long columns where the surface runs along z)
// is spread over all waves instead of being one wave's whole share. gridDim.x is a multiple of 8.
__device__ __forceinline__ size_t icp_tile_of(int tt, int tiles_per_wave) {
    const size_t per_group = (size_t)(gridDim.x >> 3) * TILE_WAVES;
    const size_t j = (size_t)(blockIdx.x >> 3) * TILE_WAVES + (threadIdx.x >> 6);
    return (size_t)(blockIdx.x & 7u) * per_group * (size_t)tiles_per_wave + (size_t)tt * per_group + j;
}

// queue: source positions left to k_icp_walk, in WALK_SHARDS parts: the tiles are cut into that many contiguous runs of
// tiles_per_shard, run r appends to the part that starts at entry 64 * tiles_per_shard * r and counts in qcount[16 * r]
// (zero before the launch; k_reduce_sums clears them for the next pass). With one counter for all, every tile's
// addition would queue at one word, which takes about 88 of them per microsecond.
// Tiles are handed out at run time (they differ several times in cost): workgroups b, b + 8, b + 16 ... run on one XCD
// (round-robin dispatch) and such a group shares one contiguous eighth of the tiles, taken in order through one of
// eight counters (64-bit, one cache line each, never reset: the host passes the value they had before this launch),
// so waves that are resident together work on neighbouring tiles, whose cell columns overlap, out of one L2. One
// counter takes about 88 additions per microsecond; an eighth of the tiles of a million points needs 2 000. pair (in / out): per source position, a copy of the target point it is paired with (see best_pair).
#ifdef ICPX_NN_STATS
__device__ unsigned long long g_nn_stats[16];
__device__ unsigned long long g_ball_t[8];   // wall-clock ticks of k_icp_ball by phase: candidates, geometry, columns, staging, compare, resolve, whole tile
__device__ unsigned long long g_nn_t0[8192], g_nn_t1[8192];
__device__ unsigned long long g_ball_t0[8192], g_ball_t1[8192], g_ball_busy[8192], g_ball_n[8192];
__device__ unsigned g_nn_tile_t[1 << 18], g_nn_tile_steps[1 << 18], g_nn_tile_total[1 << 18];
#endif
#ifdef ICPX_NN_STATS
#define BALL_T(k) do { const unsigned long long t_ = wall_clock64(); if (lane == 0) atomicAdd(&g_ball_t[k], t_ - t_last); t_last = t_; } while (0)
#else
#define BALL_T(k) do { } while (0)
#endif

// Open lanes of a tile (`pend`) go to the queues of their part of the cloud: the walk queue (entries from the front of
// the part's stretch, counted in qcount[16 part]) or, for a query SOLO_MIN_REACH to SOLO_MAX_REACH cells from the
// candidate it holds, the queue of k_icp_solo (entries from the back of the same stretch, counted in qcount[16 part + 3];
// a stretch has a slot for every lane of the part's tiles, so the two cannot meet).
__device__ __forceinline__ void icp_enqueue_open(unsigned long long pend, bool has_candidate, double d2, float h, size_t shard, unsigned tiles_per_shard,
                                                 unsigned* __restrict__ queue, unsigned* __restrict__ qcount, unsigned i, int lane) {
    const double lo = (double)SOLO_MIN_REACH * h, hi = (double)SOLO_MAX_REACH * h;
    const bool mine = (pend >> lane) & 1ull;
    const bool solo = mine && has_candidate && d2 >= lo * lo && d2 < hi * hi;
    const unsigned long long m_solo = __ballot(solo), m_walk = pend & ~m_solo;
    unsigned at_w = 0, at_s = 0;
    if (lane == 0) {
        if (m_walk) at_w = atomicAdd(&qcount[shard * 16], (unsigned)__popcll(m_walk));
        if (m_solo) at_s = atomicAdd(&qcount[shard * 16 + 3], (unsigned)__popcll(m_solo));
    }
    at_w = __shfl(at_w, 0); at_s = __shfl(at_s, 0);
    const size_t base = shard * (size_t)tiles_per_shard * 64;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (mine && !solo) queue[base + at_w + __popcll(m_walk & below)] = i;
    if (solo) queue[base + (size_t)tiles_per_shard * 64 - 1 - (at_s + __popcll(m_solo & below))] = i;
}

__global__ __launch_bounds__(64 * TILE_WAVES) void k_icp_tile(GridView g, const float4* __restrict__ src, size_t n, Mat34 T, const Mat34* __restrict__ Tdev, size_t nt,
                                                              int* __restrict__ idx_out, float4* __restrict__ pair, size_t tiles_per_group,
                                                              unsigned long long* __restrict__ counters, unsigned long long counter_base,
                                                              unsigned* __restrict__ queue, unsigned* __restrict__ qcount, unsigned tiles_per_shard,
                                                              unsigned* __restrict__ far_queue, unsigned long long* __restrict__ far_mask) {
    if (Tdev) T = *Tdev;              // the transform as the previous pass of a device-side loop left it
    __shared__ float4 sw_pts[TILE_WAVES][TILE_CAP + 8];         // as four arrays of TILE_CAP words: x, y, z, slot number
    __shared__ int sw_first[TILE_WAVES][TILE_COLS];       // per column of the box: start of its run in the sorted points
    __shared__ int sw_pre[TILE_WAVES][TILE_COLS + 4];     // ... and the number of box points in the columns before it
    float* const s_x = reinterpret_cast<float*>(sw_pts[threadIdx.x >> 6]);
    float* const s_y = s_x + TILE_CAP + 8;   // room for the padding of the last group
    float* const s_z = s_y + TILE_CAP + 8;
    float* const s_w = s_z + TILE_CAP + 8;
    int* const s_first = sw_first[threadIdx.x >> 6];
    int* const s_pre = sw_pre[threadIdx.x >> 6];
    const int lane = (int)(threadIdx.x & 63);
#ifdef ICPX_NN_STATS
    if (lane == 0 && icp_wave_id() < 8192) { g_nn_t0[icp_wave_id()] = wall_clock64(); }
#endif
    const double h = (double)g.h, ox = (double)g.minx, oy = (double)g.miny, oz = (double)g.minz;
    const unsigned group = blockIdx.x & 7u;

    for (;;) {
        unsigned long long taken = 0;
        if (lane == 0) taken = atomicAdd(&counters[group * 8], 1ull) - counter_base;
        taken = __shfl(taken, 0);
        if (taken >= tiles_per_group) break; // uniform: every wave ends here, after at most tiles_per_group + 1 additions
        const size_t tile = (size_t)group * tiles_per_group + taken, base = tile * 64;
        if (base >= n) continue;             // beyond the last tile (the counter must still see tiles_per_group + 1 additions per wave)
        const bool valid = base + lane < n;
        const size_t i = valid ? base + lane : n - 1;    // idle lanes of the last tile shadow the last point
#ifdef ICPX_NN_STATS
        const unsigned long long tile_t0 = wall_clock64();
        unsigned tile_steps = 0;
#endif
        const float4 s = src[i];
        double p[3];
        icp_transform(T, s, p);
        float4 bt;
        Best b = best_pair(pair, i, p[0], p[1], p[2], bt);
        if (b.pos < 0) cand_lookup(g, p[0], p[1], p[2], b, &bt);      // no previous pair (a first pass): the table's candidate
        const int cx = cell_coord((float)p[0], g.minx, g.inv, g.dx), cy = cell_coord((float)p[1], g.miny, g.inv, g.dy),
                  cz = cell_coord((float)p[2], g.minz, g.inv, g.dz);
        const double margin = 1e-3 * h + 1e-6 * (fabs(p[0]) + fabs(p[1]) + fabs(p[2]));
        // What a lane needs staged. With a previous pair less than TILE_NEAR cells away (the converged passes, and a growing
        // share of the points before that): its nearest point is no further than that pair, so it lies in the cells that
        // the ball of that radius around the query touches, two or three per axis, and what the lane finds among them is
        // final. Without one: its own cell and the 26 around it, and what it finds is final only if it is nearer than the
        // faces of that block. The box staged is the union of the lanes' blocks, and of a column of the box only the run
        // of cells between the lowest and the highest that some lane asked for (about half the points of a box whose
        // every column runs the full height).
        const double near2 = (double)TILE_NEAR * h * (double)TILE_NEAR * h;
        const bool near_lane = b.pos >= 0 && b.d2 < near2;
        int lx0 = max(cx - 1, 0), lx1 = min(cx + 1, g.dx - 1), ly0 = max(cy - 1, 0), ly1 = min(cy + 1, g.dy - 1), lz0 = max(cz - 1, 0), lz1 = min(cz + 1, g.dz - 1);
        if (near_lane) {
            const double rr = __dsqrt_rn(b.d2) * (1.0 + 1e-9) + 2.0 * margin;
            lx0 = cell_coord((float)(p[0] - rr), g.minx, g.inv, g.dx); lx1 = cell_coord((float)(p[0] + rr), g.minx, g.inv, g.dx);
            ly0 = cell_coord((float)(p[1] - rr), g.miny, g.inv, g.dy); ly1 = cell_coord((float)(p[1] + rr), g.miny, g.inv, g.dy);
            lz0 = cell_coord((float)(p[2] - rr), g.minz, g.inv, g.dz); lz1 = cell_coord((float)(p[2] + rr), g.minz, g.inv, g.dz);
        }
        const int bx0 = wave_min(lx0), bx1 = wave_max(lx1), by0 = wave_min(ly0), by1 = wave_max(ly1);
        const int nby = by1 - by0 + 1;
        const long long ncols_ll = (long long)(bx1 - bx0 + 1) * nby;
        const int ncols = (int)min(ncols_ll, (long long)TILE_COLS + 1);
        // squared distance from the query to the nearest place outside its own block of cells where a target point could be
        // (nothing lies beyond the grid's own faces)
        // (not needed by a near lane, whose block holds its whole ball: in the converged passes no lane computes it)
        double outside = INFINITY;
        if (!near_lane) {
            if (lx0 > 0)        { const double t = fmax(p[0] - (ox + lx0 * h) - margin, 0.0); outside = fmin(outside, t * t); }
            if (lx1 < g.dx - 1) { const double t = fmax(ox + (lx1 + 1) * h - p[0] - margin, 0.0); outside = fmin(outside, t * t); }
            if (ly0 > 0)        { const double t = fmax(p[1] - (oy + ly0 * h) - margin, 0.0); outside = fmin(outside, t * t); }
            if (ly1 < g.dy - 1) { const double t = fmax(oy + (ly1 + 1) * h - p[1] - margin, 0.0); outside = fmin(outside, t * t); }
            if (lz0 > 0)        { const double t = fmax(p[2] - (oz + lz0 * h) - margin, 0.0); outside = fmin(outside, t * t); }
            if (lz1 < g.dz - 1) { const double t = fmax(oz + (lz1 + 1) * h - p[2] - margin, 0.0); outside = fmin(outside, t * t); }
        }

        // Stage the box unless that is clearly wasted: when most lanes already hold a candidate (their previous
        // pair) and hardly any of those is near enough for the box to settle the query.
        const int seeded = __popcll(__ballot(b.pos >= 0)), promising = __popcll(__ballot(near_lane || (b.pos >= 0 && outside > b.d2)));
#ifndef ICPX_FAR_SKIP
#define ICPX_FAR_SKIP 8
#endif
        // (a tile with this many lanes far from their candidates goes to k_icp_ball whatever the box settles, and that kernel
        // takes the tile's near lanes along)
        const int far_now = __popcll(__ballot(valid && b.pos >= 0 && !(b.d2 < (double)BALL_MIN_REACH * h * (double)BALL_MIN_REACH * h)));
        bool staged = ncols <= TILE_COLS && !(seeded >= 32 && promising < 8) && !(far_queue && far_now >= ICPX_FAR_SKIP);
        int total = 0;
#ifdef ICPX_NN_STATS
        if (lane == 0) { atomicAdd(&g_nn_stats[0], 1ull); atomicAdd(&g_nn_stats[6], (unsigned long long)ncols); }
#endif
        wave_sync();                     // the previous tile is done with the column tables
        // per column of the box, the lowest and highest cell any lane asked for (in the staging area, which is free until
        // the points are copied)
        int* const s_zlo = reinterpret_cast<int*>(s_x);
        int* const s_zhi = reinterpret_cast<int*>(s_y);
        if (staged) {
            for (int c = lane; c < ncols; c += 64) { s_zlo[c] = 0x7fffffff; s_zhi[c] = -1; }
            wave_sync();
            for (int x = lx0; x <= lx1; x++)
                for (int y = ly0; y <= ly1; y++) {
                    const int c = (x - bx0) * nby + (y - by0);
                    atomicMin(&s_zlo[c], lz0); atomicMax(&s_zhi[c], lz1);
                }
            wave_sync();
        }
        if (staged) {
            for (int c0 = 0; c0 < ncols; c0 += 64) {
                const int c = c0 + lane;
                int first = 0, cnt = 0;
                if (c < ncols) {
                    const size_t cbase = ((size_t)(bx0 + c / nby) * g.dy + (by0 + c % nby)) * g.dz;
                    const int k0 = s_zlo[c], k1 = s_zhi[c];
                    if (k0 <= k1) {
                        first = g.start[cbase + k0];
                        cnt = g.start[cbase + k1 + 1] - first;
                    }
                }
                int pre = cnt;               // inclusive prefix sum over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o); if (lane >= o) pre += v; }
                if (c < ncols) { s_first[c] = first; s_pre[c] = total + pre - cnt; }
                total += __shfl(pre, 63);
            }
            if (lane == 0) s_pre[ncols] = total;
            staged = total <= TILE_MAX_PTS;
#ifdef ICPX_NN_STATS
            if (lane == 0) { atomicAdd(&g_nn_stats[1], staged ? 1ull : 0ull); atomicAdd(&g_nn_stats[7], (unsigned long long)total); }
#endif
        }
        bool ambiguous = false;
        if (staged) {
            // A branch-free float pass over the box's points, every lane against every point (read from LDS as broadcasts,
            // four points per read and coordinate). Distances are taken relative to an origin c near the tile (the first
            // lane's query, as a float): with t' = fl(t - c), w = fl(|t'|^2) (staged once per point) and q' = fl(q - c), the
            // three multiply-adds s = w - 2 q'.t' differ from |q - t|^2 - |q - c|^2 by at most 8.1 u (|t'|^2 + 2 |q'| |t'|),
            // u = 2^-24 (one rounding per coordinate of t' and q', three in w, three in s, each relative to a partial result
            // no larger than that; the result has the size of its largest term, nothing cancels). The lane keeps the smallest
            // s, its place in the list, and the second smallest. (An earlier version compared |q - t|^2 itself with the
            // point's slot in the low nine bits of the float: five to ten in a hundred lanes a few cells from the surface
            // then had two candidates too close to call, 15 000 walks per pass of a million points.)
            const float cxf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)p[0])));
            const float cyf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)p[1])));
            const float czf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)p[2])));
            const double ex = p[0] - (double)cxf, ey = p[1] - (double)cyf, ez = p[2] - (double)czf, qb2 = (ex * ex + ey * ey) + ez * ez;
            const float mx = -2.f * (float)ex, my = -2.f * (float)ey, mz = -2.f * (float)ez;
            float u1 = INFINITY, u2 = INFINITY, wmax = 0.f;
            int i1 = -1;                     // list position of the smallest
            for (int lo = 0; lo < total; lo += TILE_CAP) {
                const int hi = min(lo + TILE_CAP, total), m = hi - lo;
                wave_sync();                 // column tables written / the previous part has been compared
                // a lane per list position: its column (the last one that starts at or before it) by bisection of the prefix
                // sums, then its point (copying column by column, eight lanes a column, took three times the instructions:
                // the columns of a box hold one to five of its points each)
                for (int r0 = 0; lo + r0 < hi; r0 += 64) {
                    const int v = lo + r0 + lane;
                    int col = 0;
#pragma unroll
                    for (int step = TILE_COLS / 2; step > 0; step >>= 1) {
                        const int cand = col + step;
                        if (cand < ncols && s_pre[cand] <= v) col = cand;
                    }
                    if (v < hi) {
                        const float4 t = g.pts[s_first[col] + (v - s_pre[col])];
                        const float tx = t.x - cxf, ty = t.y - cyf, tz = t.z - czf, w = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
                        s_x[v - lo] = tx; s_y[v - lo] = ty; s_z[v - lo] = tz; s_w[v - lo] = w;
                        wmax = fmaxf(wmax, w);
                    }
                }
                // up to three far-away points fill the last group of four
                if (lane < ((4 - (m & 3)) & 3)) { s_x[m + lane] = 0.f; s_y[m + lane] = 0.f; s_z[m + lane] = 0.f; s_w[m + lane] = 3e37f; }
                wave_sync();
#ifdef ICPX_NN_STATS
                tile_steps += m;
#endif
                for (int k = 0; k < m; k += 4) {
                    const float4 X = *reinterpret_cast<const float4*>(s_x + k), Y = *reinterpret_cast<const float4*>(s_y + k),
                                 Z = *reinterpret_cast<const float4*>(s_z + k), W = *reinterpret_cast<const float4*>(s_w + k);
                    const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w}, ws[4] = {W.x, W.y, W.z, W.w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float sf = fmaf(mx, xs[j], fmaf(my, ys[j], fmaf(mz, zs[j], ws[j])));   // |t - q|^2 - |q - c|^2
                        i1 = sf < u1 ? lo + k + j : i1;
                        u2 = __builtin_amdgcn_fmed3f(u1, u2, sf);     // u1 <= u2: the second smallest of the three
                        u1 = fminf(u1, sf);
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));   // the largest |t'|^2 of the box (all lanes are here)
            // Then the exact distance of that one candidate. Every other staged point B has |q - B|^2 >= u2 + |q - c|^2 - E
            // with E the error bound above over all staged points: strictly farther when that exceeds the winner's
            // distance. Otherwise two candidates are too close to call in float and the lane is left to the walk, with
            // this candidate to start from.
            if (i1 >= 0) {
                int c = 0, hi = ncols - 1;   // the candidate's column: the last one that starts at or before it
                while (c < hi) { const int mid = (c + hi + 1) >> 1; if (s_pre[mid] <= i1) c = mid; else hi = mid - 1; }
                const int at = s_first[c] + (i1 - s_pre[c]);
                const float4 t = g.pts[at];
                const double fx = p[0] - (double)t.x, fy = p[1] - (double)t.y, fz = p[2] - (double)t.z;
                const double d = (fx * fx + fy * fy) + fz * fz;
                const double tn = (double)__fsqrt_rn(wmax) * (1.0 + 1e-6), qn = __dsqrt_rn(qb2);
                const double E = 10.0 * 0x1p-24 * (tn * tn + 2.0 * qn * tn) + 1e-30;
                ambiguous = !((double)u2 > (d - qb2) + E);
                offer_point(b, bt, d, t, at);
            }
        }
        const bool settled = staged && !ambiguous && (near_lane || outside > b.d2);
#ifdef ICPX_NN_STATS
        if (lane == 0 && tile < (1u << 18)) { g_nn_tile_t[tile] = (unsigned)(wall_clock64() - tile_t0); g_nn_tile_steps[tile] = tile_steps; g_nn_tile_total[tile] = (unsigned)total; }
#endif
        const unsigned long long pend = __ballot(valid && !settled);
#ifdef ICPX_NN_STATS
        { const unsigned long long na = __popcll(__ballot(valid && staged && ambiguous)), nu = staged ? 0ull : (unsigned long long)__popcll(pend);
          if (lane == 0) { atomicAdd(&g_nn_stats[14], na); atomicAdd(&g_nn_stats[15], nu); } }
#endif
        // open lanes without a candidate, or BALL_MIN_REACH cells and more from it (k_icp_ball's test for a far query)
        const double reach2 = (double)BALL_MIN_REACH * h * (double)BALL_MIN_REACH * h;
        const int far_lanes = __popcll(__ballot(valid && !settled && !(b.d2 < reach2)));
        if (far_queue && __popcll(pend) >= BALL_MIN_LANES && far_lanes >= 4) {
            // many of the tile's points are open and several are far from the surface: the tile goes to k_icp_ball as a whole, with the lanes it has to settle
            const size_t shard = tile / tiles_per_shard;
            if (lane == 0) {
                far_queue[shard * tiles_per_shard + atomicAdd(&qcount[shard * 16 + 1], 1u)] = (unsigned)tile;
                far_mask[tile] = pend;
            }
        } else if (pend) {
            icp_enqueue_open(pend, b.pos >= 0, b.d2, g.h, tile / tiles_per_shard, tiles_per_shard, queue, qcount, (unsigned)i, lane);
        }
        if (!valid) continue;
        if (b.pos != PAIR_POS) pair[i] = bt; // for a queued lane: the candidate its walk starts from
        if (settled && idx_out) idx_out[__float_as_int(s.w)] = b.idx;
    }
#ifdef ICPX_NN_STATS
    if (lane == 0 && icp_wave_id() < 8192) g_nn_t1[icp_wave_id()] = wall_clock64();
#endif
}

// A real target point near q, found without backtracking: down the pyramid into the first occupied child in visiting
// order, then the nearest point of the cell reached. Only an upper bound on the nearest distance is wanted.
__device__ __forceinline__ void greedy_candidate(const GridView& g, const unsigned* __restrict__ loff, double qx, double qy, double qz, Best& b) {
    int l = g.nlev - 1, ni = 0, nj = 0, nk = 0;
    const float fx = (float)qx, fy = (float)qy, fz = (float)qz;
    while (l > 0) {
        const unsigned m = g.masks[loff[l] + ((size_t)ni * lev_dim(g.dy, l) + nj) * lev_dim(g.dz, l) + nk];
        if (m == 0) return;
        const float cside = g.h * (float)(1 << (l - 1));
        const int near = (fx >= fmaf((float)(2 * ni + 1), cside, g.minx) ? 4 : 0) | (fy >= fmaf((float)(2 * nj + 1), cside, g.miny) ? 2 : 0) |
                         (fz >= fmaf((float)(2 * nk + 1), cside, g.minz) ? 1 : 0);
        const int t = __ffs(pyr_visit_order(m, near)) - 1;
        const int c = near ^ (t == 3 ? 4 : t == 4 ? 3 : t);
        ni = 2 * ni + (c >> 2); nj = 2 * nj + ((c >> 1) & 1); nk = 2 * nk + (c & 1);
        l--;
    }
    const int cell = (ni * g.dy + nj) * g.dz + nk;
    for (int k = g.start[cell]; k < g.start[cell + 1]; k++) {
        const float4 p = g.pts[k];
        const double dx = qx - (double)p.x, dy = qy - (double)p.y, dz = qz - (double)p.z;
        b.offer((dx * dx + dy * dy) + dz * dz, __float_as_int(p.w), k);
    }
}

// Tiles that k_icp_tile could not settle (queries farther from the surface than its ring of cells, or a box too full):
// one wave per tile again, but the candidates are the target points inside a ball. Every open lane holds a real target
// point as candidate (its previous pair, what the box gave, or what a greedy descent finds; the lanes then try each
// other's), at distance r from its query q, so its nearest point lies within r of q and within R = max (|q - c| + r)
// of the centre c of the queries. The ball (c, R) is covered column by column (its chord through a column of cells is
// one run of the sorted points; the columns that hold points are collected 256 at a time and their points staged in
// LDS 512 at a time), every lane against every point in float, keeping the smallest and second smallest float distance
// and the smallest one's place. The float distance is taken relative to the centre: with t' = fl(t - c), w = fl(|t'|^2)
// (staged once per point) and q' = fl(q - c), the three multiply-adds s = w - 2 q'.t' differ from |q - t|^2 - |q - c|^2
// by at most 8.1 u M, u = 2^-24, M = |t - c|^2 + 2 |q - c| |t - c| (one rounding per coordinate of t' and q', three in
// w, three in s, each relative to a partial result of size at most M): the result has the size of its largest term,
// nothing cancels. M is at most (R + 3.5 cells + spread)^2 for every staged point, so E = 10 u (...)^2 bounds the error
// of every s of the tile. Nothing inside the ball is nearer to c than the nearest candidate minus the queries'
// spread, so what is staged is a thin shell around the queries' nearest points: a few thousand points for a tile 30
// cells from the surface, where the pyramid walk expands about a hundred nodes for every single query. The winner's
// exact distance shows whether the runner-up could be as near (prune_threshold). Lanes for which it could, tiles with
// fewer than four queries BALL_MIN_REACH cells or more from their candidates (walks from a near candidate are short),
// and queries whose ball would be too wide or too full go to the walk queue. A tile whose queries lie far apart (the Morton order jumps) is cut into aligned runs of lanes, down to eight, until a run's queries lie together.
// Work is handed out at run time, like the tiles of k_icp_tile: the list of far tiles (part by part) is cut into eight
// runs, one per XCD, taken in order through the counters qcount[16 g + 2] (zero before the launch).
#ifdef ICPX_BALL_WPE
__attribute__((amdgpu_waves_per_eu(ICPX_BALL_WPE, ICPX_BALL_WPE)))
#endif
__global__ __launch_bounds__(64 * TILE_WAVES) void k_icp_ball(GridView g, const float4* __restrict__ src, size_t n, Mat34 T, const Mat34* __restrict__ Tdev, size_t nt,
                                                              int* __restrict__ idx_out, float4* __restrict__ pair, const unsigned* __restrict__ far_queue,
                                                              const unsigned long long* __restrict__ far_mask, unsigned* __restrict__ queue,
                                                              unsigned* __restrict__ qcount, unsigned tiles_per_shard, unsigned many_factor, unsigned few_tiles) {
    if (Tdev) T = *Tdev;              // the transform as the previous pass of a device-side loop left it
    __shared__ float sw_xyz[TILE_WAVES][4][TILE_CAP + 8];
    __shared__ int sw_first[TILE_WAVES][TILE_COLS];
    __shared__ int sw_pre[TILE_WAVES][TILE_COLS + 4];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    float* const s_x = sw_xyz[threadIdx.x >> 6][0];
    float* const s_y = sw_xyz[threadIdx.x >> 6][1];
    float* const s_z = sw_xyz[threadIdx.x >> 6][2];
    float* const s_w = sw_xyz[threadIdx.x >> 6][3];
    int* const s_first = sw_first[threadIdx.x >> 6];
    int* const s_pre = sw_pre[threadIdx.x >> 6];
    const int lane = (int)(threadIdx.x & 63);
#ifdef ICPX_NN_STATS
    if (lane == 0 && icp_wave_id() < 8192) { g_ball_t0[icp_wave_id()] = wall_clock64(); g_ball_busy[icp_wave_id()] = 0; g_ball_n[icp_wave_id()] = 0; }
#endif
    const unsigned my_count = qcount[lane * 16 + 1];            // lane r: far tiles of part r (WALK_SHARDS == 64)
    const unsigned most = (unsigned)wave_max((int)my_count);
    const unsigned items = most * WALK_SHARDS, group = blockIdx.x & 7u, per_group = (items + 7) / 8;
    const float h = g.h, big = 3e38f;
    // A tile with a very full or very wide ball is one long job for one wave. With a few far tiles per wave such jobs end up
    // as the kernel's tail and their points are better walked; with dozens per wave (a cloud of many millions of points far
    // from its target, whose balls hold four times the points of a million-point cloud's) they average out, and the walk
    // would take several times longer for them.
    const bool many = items >= many_factor * gridDim.x * TILE_WAVES;
    unsigned total_tiles = my_count;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total_tiles += __shfl_xor(total_tiles, o);
    const bool few = total_tiles < few_tiles;
    const int max_pts = many ? 16 * BALL_MAX_PTS : BALL_MAX_PTS, max_cols = many ? 4 * BALL_MAX_COLS : BALL_MAX_COLS;

    // A wave whose XCD's run is used up goes on with the next XCD's (and so on round the eight): the runs differ in work,
    // on the 16 M pair by a factor of 2.5 between the first XCD to finish and the last (wave end times from 45 to 121 ms
    // in the first pass); the tiles taken that way come out of another L2, but only at the end of the kernel.
#ifndef ICPX_BALL_STEAL
#define ICPX_BALL_STEAL 1
#endif
    unsigned gsel = group, hops = 0;
    for (;;) {
        unsigned taken = 0;
        if (lane == 0) {
            // a look before the addition: additions to one word go one after the other, and with little to do (the converged
            // passes) the kernel would last as long as its waves' additions take
            taken = __hip_atomic_load(&qcount[gsel * 16 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (taken < per_group) taken = atomicAdd(&qcount[gsel * 16 + 2], 1u);
        }
        taken = __shfl(taken, 0);
        const unsigned item = gsel * per_group + taken;
        if (taken >= per_group || item >= items) {            // uniform
            if (!ICPX_BALL_STEAL || ++hops >= 8) break;
            gsel = (gsel + 1) & 7u;
            continue;
        }
        const unsigned shard = item / most, e = item % most;
        if (e >= (unsigned)__shfl((int)my_count, (int)shard)) continue;
        const size_t tile = far_queue[(size_t)shard * tiles_per_shard + e], base = tile * 64;
        const unsigned long long open = far_mask[tile];
        if (few) {
            // Few far tiles (the later passes): a tile is a chain of dependent steps for one wave, 150 to 300 us whatever
            // else the GPU does, and with fewer tiles than waves that is what the kernel lasts. The open lanes go to
            // k_icp_solo instead, a wave per query, whose hundreds of short jobs fill the GPU.
            if ((open >> lane) & 1ull) {
                unsigned at = 0;
                const unsigned first = (unsigned)__ffsll((long long)open) - 1u;
                if (lane == (int)first) at = atomicAdd(&qcount[shard * 16 + 3], (unsigned)__popcll(open));
                at = __shfl(at, (int)first);
                queue[(size_t)shard * tiles_per_shard * 64 + (size_t)tiles_per_shard * 64 - 1 - (at + __popcll(open & ((1ull << lane) - 1ull)))] = (unsigned)(base + lane);
            }
            continue;
        }
#ifdef ICPX_NN_STATS
        unsigned long long t_last = wall_clock64();
        const unsigned long long t_tile = t_last;
#endif
        const bool mine = (open >> lane) & 1ull;
        const size_t i = base + lane < n ? base + lane : n - 1;
        const float4 s = src[i];
        double p[3];
        icp_transform(T, s, p);
        float4 bt;
        Best b = best_pair(pair, i, p[0], p[1], p[2], bt);
        if (mine && b.pos < 0) cand_lookup(g, p[0], p[1], p[2], b, &bt);
        const bool unpaired = __any(mine && b.pos < 0);          // no table (it is switched off): a greedy descent, then the neighbours' finds
        if (mine && b.pos < 0) {
            greedy_candidate(g, s_loff, p[0], p[1], p[2], b);
            if (b.pos >= 0) bt = g.pts[b.pos];
        }
        // The lanes' queries are neighbours: try each other's candidates. Only where some lane came without a previous pair
        // (the first pass): what a greedy descent finds can be far off, a previous pair hardly is, and the six dependent
        // reads cost a fifth of a tile's time.
#pragma unroll
        for (int o = 1; unpaired && o < 64; o <<= 1) {
            const int other = __shfl_xor(b.pos, o);
            if (mine && other >= 0 && other != PAIR_POS && other != b.pos) {       // (a lane's previous pair is not passed on)
                const float4 t = g.pts[other];
                const double dx = p[0] - (double)t.x, dy = p[1] - (double)t.y, dz = p[2] - (double)t.z;
                offer_point(b, bt, (dx * dx + dy * dy) + dz * dz, t, other);
            }
        }
        BALL_T(0);
        const float qx = (float)p[0], qy = (float)p[1], qz = (float)p[2];
        const double r_q = __dsqrt_rn(b.d2);                    // infinite without a candidate (empty target)
        const bool want = mine && r_q < 1e30;
        const bool far = want && r_q * (double)g.inv >= (double)BALL_MIN_REACH;
        bool settled = false;
        int staged_pts = 0;
        // parts: aligned runs of lanes, the whole tile first; a part whose queries lie too far apart is cut in two
        for (int off = 0, size = 64; off < 64;) {
            const bool sel = want && lane >= off && lane < off + size;
            const int here = off, width = size;
            off += size;                                        // what comes next unless this part is cut
            size = min(off & -off, 64);
            // worth a ball only for several queries away from the surface (the near ones then come along for free); a walk from a
            // near candidate is short, and eight lanes per query beat one wave per tile
            if (__popcll(__ballot(sel && far)) < 4) {
#ifdef ICPX_NN_STATS
                { const unsigned long long k = __popcll(__ballot(sel)); if (lane == 0) atomicAdd(&g_nn_stats[8], k); }
#endif
                continue;
            }
            // the ball: centre of the selected queries' box, radius over the selected lanes
            float lox = sel ? qx : big, hix = sel ? qx : -big, loy = sel ? qy : big, hiy = sel ? qy : -big, loz = sel ? qz : big, hiz = sel ? qz : -big;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lox = fminf(lox, __shfl_xor(lox, o)); hix = fmaxf(hix, __shfl_xor(hix, o));
                loy = fminf(loy, __shfl_xor(loy, o)); hiy = fmaxf(hiy, __shfl_xor(hiy, o));
                loz = fminf(loz, __shfl_xor(loz, o)); hiz = fmaxf(hiz, __shfl_xor(hiz, o));
            }
            const float cx = 0.5f * (lox + hix), cy = 0.5f * (loy + hiy), cz = 0.5f * (loz + hiz);
            // the query relative to the centre, and its squared length
            const double ex = p[0] - (double)cx, ey = p[1] - (double)cy, ez = p[2] - (double)cz, qb2 = (ex * ex + ey * ey) + ez * ez;
            const double spread_d = sel ? __dsqrt_rn(qb2) : 0.0;
            float R = sel ? (float)(spread_d + r_q) : 0.f, spread = (float)spread_d;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { R = fmaxf(R, __shfl_xor(R, o)); spread = fmaxf(spread, __shfl_xor(spread, o)); }
            if (spread * g.inv > BALL_MAX_SPREAD) {
#ifdef ICPX_NN_STATS
                { const unsigned long long k = __popcll(__ballot(sel)); if (lane == 0 && width == 8) atomicAdd(&g_nn_stats[9], k); }
#endif
                if (width > 8) { off = here; size = width / 2; }
                continue;
            }
            // a little more than the ball: covers the roundings here and points filed a hair outside their cell's box
            const float Rw = R * (1.f + 1e-5f) + 0.02f * h, R2 = Rw * Rw;
            const float reach_cells = Rw * g.inv;
            if (!(reach_cells < 4096.f)) continue;
            const int w = (int)reach_cells + 2;
            const int ccx = (int)floorf((cx - g.minx) * g.inv), ccy = (int)floorf((cy - g.miny) * g.inv);
            const int x0 = max(ccx - w, 0), x1 = min(ccx + w, g.dx - 1), y0 = max(ccy - w, 0), y1 = min(ccy + w, g.dy - 1);
            const int nby = max(y1 - y0 + 1, 1);
            const long long nc = (long long)max(x1 - x0 + 1, 0) * (long long)max(y1 - y0 + 1, 0);
#ifdef ICPX_NN_STATS
            { const unsigned long long k = __popcll(__ballot(sel)); if (lane == 0 && nc > max_cols) atomicAdd(&g_nn_stats[10], k); }
#endif
            if (nc > max_cols) continue;
            const int ncols = (int)nc;

            const float mx = -2.f * (float)ex, my = -2.f * (float)ey, mz = -2.f * (float)ez;
            float u1 = INFINITY, u2 = INFINITY;
            int best_at = -1, nfill = 0, total = 0;
            bool covered = true;
            wave_sync();                                        // whoever used the tables before is done with them
            BALL_T(1);
            constexpr int AHEAD = 8;             // rounds of 64 columns whose cell_start reads are in flight together
            int firsts[AHEAD], ends[AHEAD];
            for (int c0 = 0; c0 < ncols; c0 += 64) {
                // 64 columns: those that hold points inside the ball join the batch
                const int slot = (c0 >> 6) % AHEAD;
                if (slot == 0) {
#pragma unroll
                    for (int a = 0; a < AHEAD; a++) {
                        const int c = c0 + 64 * a + lane;
                        firsts[a] = 0; ends[a] = 0;
                        if (c < ncols) {
                            const int ci = x0 + c / nby, cj = y0 + c % nby;
                            const float lx = fmaf((float)ci, h, g.minx), ly = fmaf((float)cj, h, g.miny);
                            const float dx = fmaxf(fmaxf(lx - cx, cx - (lx + h)), 0.f), dy = fmaxf(fmaxf(ly - cy, cy - (ly + h)), 0.f);
                            const float rest = R2 - (dx * dx + dy * dy);
                            if (rest >= 0.f) {
                                const float half = __fsqrt_rn(rest);
                                const int k0 = max((int)floorf((cz - half - g.minz) * g.inv - 0.02f), 0);
                                const int k1 = min((int)floorf((cz + half - g.minz) * g.inv + 0.02f), g.dz - 1);
                                if (k0 <= k1) {
                                    const size_t cbase = ((size_t)ci * g.dy + cj) * g.dz;
                                    firsts[a] = g.start[cbase + k0];
                                    ends[a] = g.start[cbase + k1 + 1];
                                }
                            }
                        }
                    }
                }
                int first = 0, cnt = 0;
#pragma unroll
                for (int a = 0; a < AHEAD; a++) if (slot == a) { first = firsts[a]; cnt = ends[a] - firsts[a]; }
                const unsigned long long has = __ballot(cnt > 0);
                if (has) {
                    int pre = cnt;               // inclusive prefix sum over the lanes
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o); if (lane >= o) pre += v; }
                    if (cnt > 0) {
                        const int at = nfill + __popcll(has & ((1ull << lane) - 1ull));
                        s_first[at] = first; s_pre[at] = total + pre - cnt;
                    }
                    nfill += __popcll(has);
                    total += __shfl(pre, 63);
                }
                const bool last = c0 + 64 >= ncols;
                if (nfill == 0 || (!last && nfill <= TILE_COLS - 64)) continue;
                // the batch: nfill columns, total points
                BALL_T(2);
                staged_pts += total;
                if (staged_pts > max_pts) { covered = false; break; }
                if (lane == 0) s_pre[nfill] = total;
                int i1 = -1;                     // list position of the smallest distance, if it is found in this batch
                for (int lo = 0; lo < total; lo += TILE_CAP) {
                    const int hi = min(lo + TILE_CAP, total), m = hi - lo;
                    wave_sync();                 // column tables written / the previous part has been compared
                    // a lane per list position: its column by bisection of the prefix sums, then all of the lane's reads at once
                    // (column by column, as k_icp_tile copies its few dozen columns, the reads of the next column would wait for
                    // the LDS writes of this one: a ball has hundreds of short columns)
                    // (the eight bisections of a lane advance in step, so that their LDS reads are in flight together)
                    int src_at[TILE_CAP / 64], col[TILE_CAP / 64];
#pragma unroll
                    for (int r = 0; r < TILE_CAP / 64; r++) col[r] = 0;
#pragma unroll
                    for (int step = TILE_COLS / 2; step > 0; step >>= 1) {
#pragma unroll
                        for (int r = 0; r < TILE_CAP / 64; r++) {
                            const int cand = col[r] + step;
                            if (cand < nfill && s_pre[cand] <= lo + 64 * r + lane) col[r] = cand;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < TILE_CAP / 64; r++) {
                        const int v = lo + 64 * r + lane;
                        src_at[r] = v < hi ? s_first[col[r]] + (v - s_pre[col[r]]) : -1;
                    }
                    float4 pt[TILE_CAP / 64];
#pragma unroll
                    for (int r = 0; r < TILE_CAP / 64; r++) pt[r] = src_at[r] >= 0 ? g.pts[src_at[r]] : make_float4(3e18f, 3e18f, 3e18f, 0.f);
#pragma unroll
                    for (int r = 0; r < TILE_CAP / 64; r++)
                        if (src_at[r] >= 0) {        // the point relative to the centre, and its squared length
                            const float tx = pt[r].x - cx, ty = pt[r].y - cy, tz = pt[r].z - cz;
                            s_x[64 * r + lane] = tx; s_y[64 * r + lane] = ty; s_z[64 * r + lane] = tz;
                            s_w[64 * r + lane] = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
                        }
                    if (lane < ((4 - (m & 3)) & 3)) { s_x[m + lane] = 0.f; s_y[m + lane] = 0.f; s_z[m + lane] = 0.f; s_w[m + lane] = 3e37f; }   // fills the last group of four
                    wave_sync();
                    BALL_T(3);
                    for (int k = 0; k < m; k += 4) {
                        const float4 X = *reinterpret_cast<const float4*>(s_x + k), Y = *reinterpret_cast<const float4*>(s_y + k),
                                     Z = *reinterpret_cast<const float4*>(s_z + k), W = *reinterpret_cast<const float4*>(s_w + k);
                        const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w}, ws[4] = {W.x, W.y, W.z, W.w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float sf = fmaf(mx, xs[j], fmaf(my, ys[j], fmaf(mz, zs[j], ws[j])));   // |t - q|^2 - |q|^2, both relative to the centre
                            i1 = sf < u1 ? lo + k + j : i1;
                            u2 = __builtin_amdgcn_fmed3f(u1, u2, sf);     // u1 <= u2: the second smallest of the three
                            u1 = fminf(u1, sf);
                        }
                    }
                    BALL_T(4);
                }
                if (i1 >= 0) {                   // the column of that list position: the last one that starts at or before it
                    int cc = 0, hi = nfill - 1;
                    while (cc < hi) { const int mid = (cc + hi + 1) >> 1; if (s_pre[mid] <= i1) cc = mid; else hi = mid - 1; }
                    best_at = s_first[cc] + (i1 - s_pre[cc]);
                }
                nfill = 0; total = 0;
                wave_sync();                     // the tables are free again
                BALL_T(5);
            }
#ifdef ICPX_NN_STATS
            { const unsigned long long k = __popcll(__ballot(sel)); if (lane == 0 && !covered) atomicAdd(&g_nn_stats[11], k); }
#endif
            if (sel && covered && best_at >= 0) {
                const float4 t = g.pts[best_at];
                const double ex = p[0] - (double)t.x, ey = p[1] - (double)t.y, ez = p[2] - (double)t.z;
                const double d = (ex * ex + ey * ey) + ez * ez;
                // every other staged point B has |q - B|^2 >= u2 + |q - c|^2 - E (see above): strictly farther when that exceeds d
                const double far_most = (double)Rw + 3.5 * (double)h + (double)spread * (1.0 + 1e-5), E = 10.0 * 0x1p-24 * far_most * far_most;
                settled = (double)u2 > (d - qb2) + E;
                offer_point(b, bt, d, t, best_at);
            }
        }
        const unsigned long long pend = __ballot(mine && !settled);
#ifdef ICPX_NN_STATS
        if (lane == 0) atomicAdd(&g_ball_t[6], wall_clock64() - t_tile);
        if (lane == 0 && icp_wave_id() < 8192) { g_ball_busy[icp_wave_id()] += wall_clock64() - t_tile; g_ball_n[icp_wave_id()]++; g_ball_t1[icp_wave_id()] = wall_clock64(); }
        const unsigned long long k_near = __popcll(__ballot(mine && !want));
        if (lane == 0) { atomicAdd(&g_nn_stats[2], 1ull); atomicAdd(&g_nn_stats[3], __popcll(pend) < __popcll(open) ? 1ull : 0ull); atomicAdd(&g_nn_stats[4], (unsigned long long)staged_pts);
                         atomicAdd(&g_nn_stats[5], (unsigned long long)__popcll(pend)); atomicAdd(&g_nn_stats[12], k_near); }
#endif
        if (pend) icp_enqueue_open(pend, b.pos >= 0, b.d2, g.h, shard, tiles_per_shard, queue, qcount, (unsigned)i, lane);
        if (!mine) continue;
        if (b.pos != PAIR_POS) pair[i] = bt;
        if (settled && idx_out) idx_out[__float_as_int(s.w)] = b.idx;
    }
}

// The queued source points: eight lanes each, the pyramid walk of nn_search8 from the candidate in pair[]. The work is
// cut into groups of NN_OCTETS consecutive entries of one part of the queue (see k_icp_tile), numbered part by part,
// and dealt to the workgroups in turn: the workgroups that are resident together work on one stretch of the cloud
// (whose cells and pyramid nodes then stay in L2: spreading them over the whole cloud cost a third more time), and
// over its life every workgroup gets groups from all over the cloud (how long a walk takes depends on where the query is).
__global__ __launch_bounds__(256) void k_icp_walk(GridView g, const float4* __restrict__ src, Mat34 T, const Mat34* __restrict__ Tdev, size_t nt, const unsigned* __restrict__ queue,
                                                  const unsigned* __restrict__ qcount, unsigned tiles_per_shard, int* __restrict__ idx_out,
                                                  float4* __restrict__ pair, size_t n_all) {
    if (Tdev) T = *Tdev;              // the transform as the previous pass of a device-side loop left it
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ float slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __shared__ unsigned s_count[WALK_SHARDS];
    if (threadIdx.x < WALK_SHARDS) s_count[threadIdx.x] = qcount[threadIdx.x * 16];
    __syncthreads();
    if (n_all) {
        // no queue: every source point is walked (small clouds, see icp_step_enqueue)
        for (size_t i = (size_t)blockIdx.x * NN_OCTETS + (threadIdx.x >> 3); i < n_all; i += (size_t)gridDim.x * NN_OCTETS) {
            const float4 s = src[i];
            double p[3];
            icp_transform(T, s, p);
            float4 bt;
            Best b0 = best_pair(pair, i, p[0], p[1], p[2], bt);
            if (b0.pos < 0) cand_lookup(g, p[0], p[1], p[2], b0, &bt);
            const Best b = nn_search8(g, p[0], p[1], p[2], snode, slb, (int)(threadIdx.x >> 3), NN_OCTETS, s_loff, b0);
            if ((threadIdx.x & 7) != 0) continue;
            if (b.pos != b0.pos) bt = g.pts[b.pos];           // (the walk only offers points by their position)
            if (b.pos != PAIR_POS) pair[i] = bt;
            if (idx_out) idx_out[__float_as_int(s.w)] = b.idx;
        }
        return;
    }
    unsigned most = 0;
    for (int r = 0; r < WALK_SHARDS; r++) most = max(most, s_count[r]);
#ifdef ICPX_NN_STATS
    if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long w = 0; for (int r = 0; r < WALK_SHARDS; r++) w += s_count[r]; atomicAdd(&g_nn_stats[13], w); }
#endif
    const size_t per_part = (most + NN_OCTETS - 1) / NN_OCTETS, items = per_part * WALK_SHARDS;
    for (size_t item = blockIdx.x; item < items; item += gridDim.x) {
        const unsigned shard = (unsigned)(item / per_part), e = (unsigned)(item % per_part) * NN_OCTETS + (threadIdx.x >> 3);
        if (e >= s_count[shard]) continue;
        const size_t i = queue[(size_t)shard * tiles_per_shard * 64 + e];
        const float4 s = src[i];
        double p[3];
        icp_transform(T, s, p);
        float4 bt;
        const Best b0 = best_pair(pair, i, p[0], p[1], p[2], bt);
        const Best b = nn_search8(g, p[0], p[1], p[2], snode, slb, (int)(threadIdx.x >> 3), NN_OCTETS, s_loff, b0);
        if ((threadIdx.x & 7) != 0) continue;
        if (b.pos != b0.pos) pair[i] = g.pts[b.pos];
        if (idx_out) idx_out[__float_as_int(s.w)] = b.idx;
    }
}

// The queries k_icp_tile and k_icp_ball left open that are several cells from their candidate: a wave each. Such a query's
// pyramid walk expands a hundred nodes and more, one after the other, each a chain of dependent reads: the walk kernel
// then lasts as long as its longest walk (half a millisecond in the early passes of a million points) with most of the
// GPU idle. Here the 64 lanes share the ball around the query whose radius is the candidate's distance: lane l takes
// columns l, l + 64, ... of the cells under the ball, reads the two ends of the column's chord from cell_start and scans
// the chord's points (float distance first, the exact one for points the float value cannot rule out); the lanes'
// results are reduced by (distance, index). Every point as near as the candidate lies in the ball, so the result is the
// exact nearest point with the lowest index, as from the walk.
__global__ __launch_bounds__(256) void k_icp_solo(GridView g, const float4* __restrict__ src, Mat34 T, const Mat34* __restrict__ Tdev, size_t nt, const unsigned* __restrict__ queue,
                                                  const unsigned* __restrict__ qcount, unsigned tiles_per_shard, int* __restrict__ idx_out, float4* __restrict__ pair) {
    if (Tdev) T = *Tdev;              // the transform as the previous pass of a device-side loop left it
    __shared__ unsigned s_count[WALK_SHARDS];
    if (threadIdx.x < WALK_SHARDS) s_count[threadIdx.x] = qcount[threadIdx.x * 16 + 3];
    __syncthreads();
    unsigned most = 0;
    for (int r = 0; r < WALK_SHARDS; r++) most = max(most, s_count[r]);
    const int lane = (int)(threadIdx.x & 63);
    const size_t nwaves = (size_t)gridDim.x * 4, items = (size_t)most * WALK_SHARDS;
    const float h = g.h;
    // items are numbered part by part; the four waves of a workgroup take four consecutive ones
    for (size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < items; item += nwaves) {
        const unsigned shard = (unsigned)(item / most), e = (unsigned)(item % most);
        if (e >= s_count[shard]) continue;
        const size_t i = queue[(size_t)shard * tiles_per_shard * 64 + (size_t)tiles_per_shard * 64 - 1 - e];
        const float4 s = src[i];
        double p[3];
        icp_transform(T, s, p);
        float4 bt;
        Best b = best_pair(pair, i, p[0], p[1], p[2], bt);        // the same in every lane
        if (b.pos < 0) cand_lookup(g, p[0], p[1], p[2], b, &bt);
        if (b.pos < 0) {                                            // no candidate anywhere: an empty target (the pair stays "none")
            if (lane == 0 && idx_out) idx_out[__float_as_int(s.w)] = b.idx;
            continue;
        }
        const int pos0 = b.pos;
        const QueryF f(p[0], p[1], p[2], (double)h);
        const float r = (float)__dsqrt_rn(b.d2);
        const float Rw = r * (1.f + 1e-5f) + 0.02f * h, R2 = Rw * Rw;
        const float qx = f.x, qy = f.y, qz = f.z;
        const int w = (int)(Rw * g.inv) + 2;
        const int ccx = (int)floorf((qx - g.minx) * g.inv), ccy = (int)floorf((qy - g.miny) * g.inv);
        const int x0 = max(ccx - w, 0), x1 = min(ccx + w, g.dx - 1), y0 = max(ccy - w, 0), y1 = min(ccy + w, g.dy - 1);
        const int nby = max(y1 - y0 + 1, 1);
        const int ncols = max(x1 - x0 + 1, 0) * max(y1 - y0 + 1, 0);
        float thr = f.point_threshold(b.d2);
        constexpr int AHEAD = 4;            // columns per lane whose cell_start reads are in flight together
        for (int c0 = 0; c0 < ncols; c0 += 64 * AHEAD) {
            int firsts[AHEAD], ends[AHEAD];
#pragma unroll
            for (int a = 0; a < AHEAD; a++) {
                const int c = c0 + 64 * a + lane;
                firsts[a] = 0; ends[a] = 0;
                if (c < ncols) {
                    const int ci = x0 + c / nby, cj = y0 + c % nby;
                    const float lx = fmaf((float)ci, h, g.minx), ly = fmaf((float)cj, h, g.miny);
                    const float dx = fmaxf(fmaxf(lx - qx, qx - (lx + h)), 0.f), dy = fmaxf(fmaxf(ly - qy, qy - (ly + h)), 0.f);
                    const float rest = R2 - (dx * dx + dy * dy);
                    if (rest >= 0.f) {
                        const float half = __fsqrt_rn(rest);
                        const int k0 = max((int)floorf((qz - half - g.minz) * g.inv - 0.02f), 0);
                        const int k1 = min((int)floorf((qz + half - g.minz) * g.inv + 0.02f), g.dz - 1);
                        if (k0 <= k1) {
                            const size_t cbase = ((size_t)ci * g.dy + cj) * g.dz;
                            firsts[a] = g.start[cbase + k0];
                            ends[a] = g.start[cbase + k1 + 1];
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < AHEAD; a++)
                for (int k = firsts[a]; k < ends[a]; k++) {
                    // (asking for the first four points of every column together, before knowing which columns hold any, made
                    // the kernel a fifth slower: most chords are empty)
                    const float4 t = g.pts[k];
                    const float fx = qx - t.x, fy = qy - t.y, fz = qz - t.z;
                    const float sf = (fx * fx + fy * fy) + fz * fz;
                    if (sf > thr) continue;
                    const double ex = p[0] - (double)t.x, ey = p[1] - (double)t.y, ez = p[2] - (double)t.z;
                    b.offer((ex * ex + ey * ey) + ez * ez, __float_as_int(t.w), k);
                    thr = f.point_threshold(b.d2);
                }
        }
        // minimum by (distance, index) over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double d = __shfl_xor(b.d2, o);
            const int id = __shfl_xor(b.idx, o), at = __shfl_xor(b.pos, o);
            if (d < b.d2 || (d == b.d2 && id < b.idx)) { b.d2 = d; b.idx = id; b.pos = at; }
        }
        if (lane == 0) {
            if (b.pos != pos0) bt = g.pts[b.pos];               // (the scan offers points by their position)
            if (b.pos != PAIR_POS) pair[i] = bt;
            if (idx_out) idx_out[__float_as_int(s.w)] = b.idx;
        }
    }
}

// Block-partial sums {p, q, p q^T, |p-q|^2} over the pairs in pair[]: lane l of a block adds the terms of point l of
// its tiles in tile order and the lanes are summed by a fixed butterfly, so the sums depend on the pairs alone (not on
// which points were settled by k_icp_tile and which by the walk, which varies with the previous pass).
__global__ __launch_bounds__(64 * TILE_WAVES) void k_icp_sums(GridView g, const float4* __restrict__ src, size_t n, Mat34 T, const Mat34* __restrict__ Tdev, size_t nt,
                                                 const float4* __restrict__ pair, int tiles_per_block, double* __restrict__ partials) {
    if (Tdev) T = *Tdev;              // the transform as the previous pass of a device-side loop left it
    const int lane = (int)(threadIdx.x & 63);
    double acc[ICPX_NSUMS];
#pragma unroll
    for (int k = 0; k < ICPX_NSUMS; k++) acc[k] = 0.0;
    for (int tt = 0; tt < tiles_per_block; tt++) {
        const size_t i = icp_tile_of(tt, tiles_per_block) * 64 + lane;
        if (i >= n) continue;
        double p[3];
        icp_transform(T, src[i], p);
        float4 q = pair[i];
        // no candidate at all (a NaN in T): the sums are NaN anyway
        if (__float_as_int(q.w) < 0) q = make_float4(0.f, 0.f, 0.f, 0.f);
        const double dx = p[0] - (double)q.x, dy = p[1] - (double)q.y, dz = p[2] - (double)q.z;
        icp_accumulate(acc, p, q, (dx * dx + dy * dy) + dz * dz);
    }
    icp_write_partials(acc, partials, lane);
}

// Trimmed ICP pass, first half: p = T * [src;1] and its nearest target point for every source point, stored
// under the caller's source index o: the squared distance as a sort key (the bits of a non-negative double
// order like the value), o itself as the sort value, p and the correspondence.
__global__ __launch_bounds__(256) void k_icp_pairs(GridView g, const float4* __restrict__ src, size_t n, Mat34 T,
                                                   unsigned long long* __restrict__ keys, unsigned* __restrict__ vals,
                                                   double* __restrict__ pbuf, int* __restrict__ idx, size_t nt, int* __restrict__ seed) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ float slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;   // eight lanes per source point
    if (i >= n) return;
    const float4 s = src[i];
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
    const Best b = nn_search8(g, p[0], p[1], p[2], snode, slb, (int)(threadIdx.x >> 3), NN_OCTETS, s_loff, best_seed(g, seed, i, nt, p[0], p[1], p[2]));
    if ((threadIdx.x & 7) != 0) return;
    if (seed) seed[i] = b.pos;
    const size_t o = (size_t)__float_as_int(s.w);
    keys[o] = (unsigned long long)__double_as_longlong(b.d2);
    vals[o] = (unsigned)o;
    pbuf[3 * o] = p[0]; pbuf[3 * o + 1] = p[1]; pbuf[3 * o + 2] = p[2];
    idx[o] = b.idx;
}

// Second half: block-partial sums {p, q, p q^T, |p-q|^2} over the `keep` pairs with the smallest squared
// distances (order[0 .. keep) from the stable sort: ties go to the lower source index).
__global__ __launch_bounds__(256) void k_icp_trim_sums(const unsigned* __restrict__ order, const unsigned long long* __restrict__ keys_sorted,
                                                       size_t keep, const double* __restrict__ pbuf, const int* __restrict__ idx,
                                                       const float* __restrict__ tgt_xyz, double* __restrict__ partials) {
    double acc[ICPX_NSUMS];
#pragma unroll
    for (int k = 0; k < ICPX_NSUMS; k++) acc[k] = 0.0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < keep; r += (size_t)gridDim.x * blockDim.x) {
        const size_t o = order[r];
        const double p0 = pbuf[3 * o], p1 = pbuf[3 * o + 1], p2 = pbuf[3 * o + 2];
        const size_t j = idx[o] == 0x7fffffff ? 0 : (size_t)idx[o];   // no candidate (NaN transform): see k_icp_tile
        const double q0 = (double)tgt_xyz[3 * j], q1 = (double)tgt_xyz[3 * j + 1], q2 = (double)tgt_xyz[3 * j + 2];
        acc[0] += p0; acc[1] += p1; acc[2] += p2;
        acc[3] += q0; acc[4] += q1; acc[5] += q2;
        acc[6] += p0 * q0; acc[7] += p0 * q1; acc[8] += p0 * q2;
        acc[9] += p1 * q0; acc[10] += p1 * q1; acc[11] += p1 * q2;
        acc[12] += p2 * q0; acc[13] += p2 * q1; acc[14] += p2 * q2;
        acc[15] += __longlong_as_double((long long)keys_sorted[r]);
    }
    __shared__ double sm[4][ICPX_NSUMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < ICPX_NSUMS; k++) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) sm[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < ICPX_NSUMS)
        partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// clear16: nclear groups of four words, 16 words apart, set to zero on the way (nullptr: none).
// Fixed-order final reduction of the block partials (component-major: partials[component * nblocks + block]): one
// workgroup per component, thread t adds blocks t, t + 256, ... in that order, then a tree over the threads.
__global__ __launch_bounds__(256) void k_reduce_sums(const double* __restrict__ partials, int nblocks, double* __restrict__ sums,
                                                     unsigned* __restrict__ clear16 = nullptr, int nclear = 0) {
    __shared__ double sm[256];
    if (clear16 && blockIdx.x == 0 && (int)threadIdx.x < nclear) {   // the queues' counters and the hand-out counters of k_icp_ball, for the next pass
        clear16[threadIdx.x * 16] = 0u; clear16[threadIdx.x * 16 + 1] = 0u; clear16[threadIdx.x * 16 + 2] = 0u; clear16[threadIdx.x * 16 + 3] = 0u;
    }
    const double* row = partials + (size_t)blockIdx.x * nblocks;
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) v += row[b];
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = sm[0];
}

GridView view_of(const icpx_ctx* ctx) {
    GridView g;
    g.dx = ctx->gdim[0]; g.dy = ctx->gdim[1]; g.dz = ctx->gdim[2];
    g.minx = ctx->gmin[0]; g.miny = ctx->gmin[1]; g.minz = ctx->gmin[2];
    g.h = ctx->gh; g.inv = ctx->ginv;
    g.start = ctx->cell_start.as<int>();
    g.pts = ctx->sorted_pts.as<float4>();
    g.masks = ctx->pyr_masks.as<unsigned char>();
    g.loff = ctx->pyr_tab.as<unsigned>();
    g.nlev = ctx->pyr_nlev;
    g.cand = ctx->cand_ready ? ctx->cand_tab.as<int>() : nullptr;
    g.cdx = ctx->cand_dim[0]; g.cdy = ctx->cand_dim[1]; g.cdz = ctx->cand_dim[2]; g.cshift = ctx->cand_shift;
    return g;
}

}  // namespace

// Builds the search grid over the staged target (ctx->sorted_pts holds the points in caller order).
// Cell size: start from one point per cell by volume, then — point sets sampled from surfaces fill far
// fewer cells than that — refine once so that an occupied cell holds about four points.
int icpx_grid_build(icpx_ctx* ctx) {
    const size_t n = ctx->nt;
    const double ext[3] = {(double)ctx->bbmax[0] - ctx->bbmin[0], (double)ctx->bbmax[1] - ctx->bbmin[1], (double)ctx->bbmax[2] - ctx->bbmin[2]};
    const double emax = std::max(ext[0], std::max(ext[1], ext[2]));
    const double floor_e = emax > 0 ? emax * 1e-3 : 1.0;   // degenerate (flat) extents still get one cell
    double vol = 1.0;
    for (int c = 0; c < 3; c++) vol *= std::max(ext[c], floor_e);
    double h = std::cbrt(vol / (double)n);
    // At most 1024 cells per axis (the walk packs node coordinates in ten bits), so at most 2^30 cells: 4 GiB each for the
    // counts and the starts, reached only by clouds of ten million points and more (16 M surface points: 1024^3 cells, about
    // six points per occupied cell; a cap of 3.2e7 cells left 65 per cell there and every query to the pyramid walk).
    const double max_cells = 1.08e9;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    ICPX_TRY(icpx_sort_reserve(ctx, n));
    ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * sizeof(float4)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_counters, 8 * sizeof(unsigned long long)));
    // keep the caller-order copy; sorted_pts is rewritten by the gather
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, ctx->sorted_pts.p, n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    for (int attempt = 0; attempt < 2; attempt++) {
        int dim[3];
        // No axis may need more than 1024 cells: coordinates beyond the last cell would be filed under it
        // while lying outside its box, and the search's bounds rely on the boxes.
        h = std::max(h, emax / 1023.0);
        for (;;) {
            double cells = 1.0;
            for (int c = 0; c < 3; c++) { dim[c] = (int)std::min(1024.0, std::floor(ext[c] / h) + 1.0); cells *= dim[c]; }
            if (cells <= max_cells) break;
            h *= 1.26;
        }
        ctx->gh = (float)h; ctx->ginv = 1.0f / ctx->gh;
        for (int c = 0; c < 3; c++) { ctx->gdim[c] = dim[c]; ctx->gmin[c] = ctx->bbmin[c]; }
        const size_t ncells = (size_t)dim[0] * dim[1] * dim[2];
        ICPX_TRY(icpx_reserve(ctx, ctx->cell_start, (ncells + 1) * sizeof(int)));
        ICPX_TRY(icpx_reserve(ctx, ctx->cell_count, (ncells + 1) * sizeof(int)));
        ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->cell_count.p, 0, (ncells + 1) * sizeof(int), ctx->stream));
        unsigned* distinct = reinterpret_cast<unsigned*>(ctx->edt_counters.as<unsigned long long>() + 3);
        ICPX_CHECK_HIP(ctx, hipMemsetAsync(distinct, 0, sizeof(unsigned), ctx->stream));
        GridView g = view_of(ctx);
        hipLaunchKernelGGL(k_cell_keys, dim3(blocks), dim3(256), 0, ctx->stream, ctx->scratch_q.as<float4>(), n, g,
                           ctx->sort_keys_a.as<unsigned>(), ctx->sort_vals_a.as<unsigned>());
        ICPX_TRY(icpx_launch_check(ctx, "k_cell_keys"));
        unsigned bits = 1;
        while ((1ull << bits) < ncells) bits++;
        ICPX_TRY(icpx_sort_pairs(ctx, n, bits));
        hipLaunchKernelGGL(k_cell_counts, dim3(blocks), dim3(256), 0, ctx->stream, ctx->sort_keys_b.as<unsigned>(), n,
                           ctx->cell_count.as<int>(), distinct);
        ICPX_TRY(icpx_launch_check(ctx, "k_cell_counts"));
        {   // cell_start = exclusive prefix sum of the counts (the entry behind the last cell is the number of points)
            size_t tmp = 0;
            ICPX_CHECK_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp, ctx->cell_count.as<int>(), ctx->cell_start.as<int>(), 0, ncells + 1,
                                                        rocprim::plus<int>(), ctx->stream));
            ICPX_TRY(icpx_reserve(ctx, ctx->sort_tmp, tmp));
            ICPX_CHECK_HIP(ctx, rocprim::exclusive_scan(ctx->sort_tmp.p, tmp, ctx->cell_count.as<int>(), ctx->cell_start.as<int>(), 0, ncells + 1,
                                                        rocprim::plus<int>(), ctx->stream));
        }
        unsigned occupied = 0;
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(&occupied, distinct, sizeof occupied, hipMemcpyDeviceToHost, ctx->stream));
        ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double per_cell = occupied ? (double)n / occupied : 1.0;
        if (attempt == 0 && per_cell > 6.0) { h *= std::sqrt(4.0 / per_cell); continue; }
        break;
    }
    hipLaunchKernelGGL(k_gather_pts, dim3(blocks), dim3(256), 0, ctx->stream, ctx->scratch_q.as<float4>(), ctx->sort_vals_b.as<unsigned>(), n,
                       ctx->sorted_pts.as<float4>());
    ICPX_TRY(icpx_launch_check(ctx, "k_gather_pts"));
    // occupancy pyramid over the cells
    {
        int d[3] = {ctx->gdim[0], ctx->gdim[1], ctx->gdim[2]};
        int nlev = 1;
        unsigned off = 0;
        ctx->pyr_dim[0][0] = d[0]; ctx->pyr_dim[0][1] = d[1]; ctx->pyr_dim[0][2] = d[2]; ctx->pyr_off[0] = 0;
        while ((d[0] > 1 || d[1] > 1 || d[2] > 1) && nlev < MAXLEV) {   // dims <= 1024: ten halvings reach the root
            for (int c = 0; c < 3; c++) d[c] = (d[c] + 1) / 2;
            ctx->pyr_dim[nlev][0] = d[0]; ctx->pyr_dim[nlev][1] = d[1]; ctx->pyr_dim[nlev][2] = d[2];
            ctx->pyr_off[nlev] = off;
            off += ((unsigned)d[0] * d[1] * d[2] + 3u) & ~3u;            // levels start on a word (the mask bits are set with word atomics)
            nlev++;
        }
        for (int l = nlev; l < MAXLEV; l++) { ctx->pyr_dim[l][0] = ctx->pyr_dim[l][1] = ctx->pyr_dim[l][2] = 0; ctx->pyr_off[l] = 0; }
        ctx->pyr_nlev = nlev;
        ICPX_TRY(icpx_reserve(ctx, ctx->pyr_tab, MAXLEV * sizeof(unsigned)));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->pyr_tab.p, ctx->pyr_off, MAXLEV * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
        ICPX_TRY(icpx_reserve(ctx, ctx->pyr_masks, off ? off : 4));
        ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->pyr_masks.p, 0, off ? off : 4, ctx->stream));
        unsigned char* fl = ctx->pyr_masks.as<unsigned char>();
        for (int l = 1; l < nlev; l++) {
            const int* lo = ctx->pyr_dim[l - 1]; const int* up = ctx->pyr_dim[l];
            const size_t cells = (size_t)lo[0] * lo[1] * lo[2];
            const unsigned nb = (unsigned)((cells + 255) / 256);
            unsigned* upw = reinterpret_cast<unsigned*>(fl + ctx->pyr_off[l]);
            if (l == 1)
                hipLaunchKernelGGL(k_pyr_from_cells, dim3(nb), dim3(256), 0, ctx->stream, ctx->cell_start.as<int>(), lo[0], lo[1], lo[2], upw, up[1], up[2]);
            else
                hipLaunchKernelGGL(k_pyr_from_masks, dim3(nb), dim3(256), 0, ctx->stream, fl + ctx->pyr_off[l - 1], lo[0], lo[1], lo[2], upw, up[1], up[2]);
        }
        ICPX_TRY(icpx_launch_check(ctx, "pyramid kernels"));
    }
    // candidate table (see k_cand_seed): blocks of 2^shift cells per axis, the smallest shift that keeps the table
    // below two million entries (a million points: 254^3 cells, blocks of 2, 2.1 M / 8 ... )
    ctx->cand_ready = false;
    if (ctx->cand_enable) {
        int shift = 1;
        for (;;) {
            size_t n = 1;
            for (int c = 0; c < 3; c++) { ctx->cand_dim[c] = (ctx->gdim[c] + (1 << shift) - 1) >> shift; n *= (size_t)ctx->cand_dim[c]; }
            if (n <= (size_t)ctx->cand_max_entries || shift >= 8) break;
            shift++;
        }
        ctx->cand_shift = shift;
        const size_t n = (size_t)ctx->cand_dim[0] * ctx->cand_dim[1] * ctx->cand_dim[2];
        ICPX_TRY(icpx_reserve(ctx, ctx->cand_tab, n * sizeof(int)));
        ICPX_TRY(icpx_reserve(ctx, ctx->cand_tmp, n * sizeof(int)));
        GridView g = view_of(ctx);
        const unsigned nb = (unsigned)((n + 255) / 256);
        int* a = ctx->cand_tab.as<int>(); int* b = ctx->cand_tmp.as<int>();
        hipLaunchKernelGGL(k_cand_seed, dim3(nb), dim3(256), 0, ctx->stream, g, a);
        int top = 1;
        while (top * 2 < std::max(ctx->cand_dim[0], std::max(ctx->cand_dim[1], ctx->cand_dim[2]))) top *= 2;
        int rounds = 0;
        for (int step = top; step >= 1; step >>= 1, rounds++) { hipLaunchKernelGGL(k_cand_flood, dim3(nb), dim3(256), 0, ctx->stream, g, (const int*)a, b, step); std::swap(a, b); }
        hipLaunchKernelGGL(k_cand_flood, dim3(nb), dim3(256), 0, ctx->stream, g, (const int*)a, b, 1); std::swap(a, b); rounds++;
        if (a != ctx->cand_tab.as<int>()) ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->cand_tab.p, a, n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        ICPX_TRY(icpx_launch_check(ctx, "candidate table kernels"));
        ctx->cand_ready = true;
    }
    ctx->grid_ready = true;
    return ICPX_OK;
}

extern "C" int icpx_nn(icpx_ctx* ctx, const double* q_xyz, size_t n, int where, int32_t* idx, double* d2) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->grid_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_nn: no target cloud (call icpx_set_target first)");
    if (n == 0) return ICPX_OK;
    if (!q_xyz) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_nn: queries are NULL");
    ICPX_TRY(icpx_bind(ctx));
    const double* dq = q_xyz; int* didx = idx; double* dd2 = d2;
    if (where == ICPX_HOST) {
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * 3 * sizeof(double)));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, q_xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dq = ctx->scratch_q.as<double>();
        if (idx) { ICPX_TRY(icpx_reserve(ctx, ctx->scratch_idx, n * sizeof(int))); didx = ctx->scratch_idx.as<int>(); }
        if (d2) { ICPX_TRY(icpx_reserve(ctx, ctx->scratch_d2, n * sizeof(double))); dd2 = ctx->scratch_d2.as<double>(); }
    }
    hipLaunchKernelGGL(k_nn, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, ctx->stream, view_of(ctx), dq, n, didx, dd2);
    ICPX_TRY(icpx_launch_check(ctx, "k_nn"));
    if (where == ICPX_HOST) {
        if (idx) ICPX_CHECK_HIP(ctx, hipMemcpyAsync(idx, didx, n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        if (d2) ICPX_CHECK_HIP(ctx, hipMemcpyAsync(d2, dd2, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

extern "C" int icpx_knn(icpx_ctx* ctx, const double* q_xyz, size_t n, int where, int k, int32_t* idx, double* d2) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->grid_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_knn: no target cloud (call icpx_set_target first)");
    if (k < 1 || k > ICPX_MAX_KNN) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_knn: k must be in [1, %d]", ICPX_MAX_KNN);
    if (n == 0) return ICPX_OK;
    if (!q_xyz) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_knn: queries are NULL");
    ICPX_TRY(icpx_bind(ctx));
    const double* dq = q_xyz; int* didx = idx; double* dd2 = d2;
    if (where == ICPX_HOST) {
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * 3 * sizeof(double)));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, q_xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dq = ctx->scratch_q.as<double>();
        if (idx) { ICPX_TRY(icpx_reserve(ctx, ctx->scratch_idx, n * k * sizeof(int))); didx = ctx->scratch_idx.as<int>(); }
        if (d2) { ICPX_TRY(icpx_reserve(ctx, ctx->scratch_d2, n * k * sizeof(double))); dd2 = ctx->scratch_d2.as<double>(); }
    }
    const dim3 grid((unsigned)((n + 31) / 32)), block(256);   // eight lanes per query
    if (k == 1) hipLaunchKernelGGL(k_knn<1>, grid, block, 0, ctx->stream, view_of(ctx), dq, n, k, didx, dd2);
    else if (k == 2) hipLaunchKernelGGL(k_knn<2>, grid, block, 0, ctx->stream, view_of(ctx), dq, n, k, didx, dd2);
    else if (k <= 4) hipLaunchKernelGGL(k_knn<4>, grid, block, 0, ctx->stream, view_of(ctx), dq, n, k, didx, dd2);
    else if (k <= 5) hipLaunchKernelGGL(k_knn<5>, grid, block, 0, ctx->stream, view_of(ctx), dq, n, k, didx, dd2);
    else hipLaunchKernelGGL(k_knn<8>, grid, block, 0, ctx->stream, view_of(ctx), dq, n, k, didx, dd2);
    ICPX_TRY(icpx_launch_check(ctx, "k_knn"));
    if (where == ICPX_HOST) {
        if (idx) ICPX_CHECK_HIP(ctx, hipMemcpyAsync(idx, didx, n * k * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        if (d2) ICPX_CHECK_HIP(ctx, hipMemcpyAsync(d2, dd2, n * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

extern "C" int icpx_nn_distance(icpx_ctx* ctx, const float* q_xyz, size_t n, int where, float* dist) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->grid_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_nn_distance: no target cloud (call icpx_set_target first)");
    if (n == 0) return ICPX_OK;
    if (!q_xyz || !dist) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_nn_distance: NULL buffer");
    ICPX_TRY(icpx_bind(ctx));
    const float* dq = q_xyz; float* dout = dist;
    if (where == ICPX_HOST) {
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * 3 * sizeof(float)));
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_d2, n * sizeof(float)));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, q_xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        dq = ctx->scratch_q.as<float>(); dout = ctx->scratch_d2.as<float>();
    }
    hipLaunchKernelGGL(k_nn_distance, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, ctx->stream, view_of(ctx), dq, n, dout);
    ICPX_TRY(icpx_launch_check(ctx, "k_nn_distance"));
    if (where == ICPX_HOST) ICPX_CHECK_HIP(ctx, hipMemcpyAsync(dist, dout, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

static int icp_step_reserve(icpx_ctx* ctx);

extern "C" int icpx_set_source(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where) {
    if (!ctx) return ICPX_EINVAL;
    if (!xyz || n == 0) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_set_source: empty source cloud");
    if (stride_bytes < 12 || stride_bytes % 4) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_set_source: stride must be a multiple of 4 and >= 12");
    if (n > 0x7fffffffu) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_set_source: more than 2^31-1 points");
    ICPX_TRY(icpx_bind(ctx));
    ctx->ns = 0;
    ICPX_TRY(icpx_reserve(ctx, ctx->src_pts, n * sizeof(float4)));
    ICPX_TRY(icpx_stage_points4(ctx, xyz, n, stride_bytes, where, ctx->src_pts.as<float4>()));
    ICPX_TRY(icpx_morton_sort4(ctx, ctx->src_pts.as<float4>(), n));
#ifndef ICPX_NO_CLUSTER
    // clouds that go through the tile kernels: compact tiles (a cloud that is walked point by point gains nothing)
    if ((n + 63) / 64 >= ctx->icp_walk_all_tiles) ICPX_TRY(icpx_cluster_tiles4(ctx, ctx->src_pts.as<float4>(), n));
#endif
    // correspondences of the last pass, by sorted position (-1 = none): the next pass starts its search from them
    ICPX_TRY(icpx_reserve(ctx, ctx->nn_seed, n * sizeof(int)));
    ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->nn_seed.p, 0xff, n * sizeof(int), ctx->stream));
    ICPX_TRY(icpx_reserve(ctx, ctx->nn_pair, n * sizeof(float4)));
    ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->nn_pair.p, 0xff, n * sizeof(float4), ctx->stream));
    ctx->ns = n;
    ICPX_TRY(icp_step_reserve(ctx));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

// Working memory of an ICP pass for the current source cloud: per-wave partial sums, the tile hand-out counters, the
// queues. Reserved when the source is set, so that the first pass allocates nothing (an allocation inside a pass stalls
// the stream for a millisecond).
static int icp_step_reserve(icpx_ctx* ctx) {
    const size_t tiles = (ctx->ns + 63) / 64;
    const int sblocks = (int)((std::min<size_t>((tiles + TILE_WAVES - 1) / TILE_WAVES, (size_t)ctx->num_cus * 8) + 7) / 8 * 8);
    ICPX_TRY(icpx_reserve(ctx, ctx->partials, (size_t)sblocks * TILE_WAVES * ICPX_NSUMS * sizeof(double)));
    if (!ctx->tile_counters.p) {
        ICPX_TRY(icpx_reserve(ctx, ctx->tile_counters, 8 * 64));
        ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->tile_counters.p, 0, 8 * 64, ctx->stream));
        ctx->tile_counter_base = 0;
    }
    const unsigned tiles_per_shard = (unsigned)((tiles + WALK_SHARDS - 1) / WALK_SHARDS);
    const size_t far_off = (size_t)WALK_SHARDS * 16 + (size_t)WALK_SHARDS * tiles_per_shard * 64;
    const size_t mask_off = (far_off + (size_t)WALK_SHARDS * tiles_per_shard + 1) / 2 * 2;
    const size_t queue_words = mask_off + 2 * (size_t)WALK_SHARDS * tiles_per_shard;
    if (ctx->walk_queue.cap < queue_words * sizeof(unsigned)) {
        ICPX_TRY(icpx_reserve(ctx, ctx->walk_queue, queue_words * sizeof(unsigned)));
        ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->walk_queue.p, 0, WALK_SHARDS * 16 * sizeof(unsigned), ctx->stream));   // later passes: k_reduce_sums
    }
    ICPX_TRY(icpx_reserve(ctx, ctx->sums, ICPX_NSUMS * sizeof(double)));
    return ICPX_OK;
}

static int icp_step_enqueue(icpx_ctx* ctx, const double T[16], double* sums_dev, int* idx_dev, const Mat34* Tdev = nullptr) {
    if (!ctx->grid_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_step: no target cloud (call icpx_set_target first)");
    if (ctx->ns == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_step: no source cloud (call icpx_set_source first)");
    if (!T && !Tdev) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step: T is NULL");
    // One wave per tile of 64 source points, TILE_WAVES waves per workgroup; 7 workgroups of the tile kernel fit a CU
    // (LDS). The workgroup counts are multiples of 8 (one share of the tiles per XCD).
    const size_t tiles = (ctx->ns + 63) / 64;
    const int blocks = (int)((std::min<size_t>((tiles + TILE_WAVES - 1) / TILE_WAVES, (size_t)ctx->num_cus * 7) + 7) / 8 * 8);
    const size_t tiles_per_group = (tiles + 7) / 8;
    const int sblocks = (int)((std::min<size_t>((tiles + TILE_WAVES - 1) / TILE_WAVES, (size_t)ctx->num_cus * 8) + 7) / 8 * 8);
    const int swaves = sblocks * TILE_WAVES;
    const int tiles_per_sblock = (int)((tiles + swaves - 1) / swaves);
    // the partial sums, the eight tile counters (64 B apart), the per-part counts of queued points and the queues (64
    // entries per tile): reserved by icpx_set_source, here only for a context whose source was set before a resize
    ICPX_TRY(icp_step_reserve(ctx));
    const unsigned tiles_per_shard = (unsigned)((tiles + WALK_SHARDS - 1) / WALK_SHARDS);
    // counters (per part: walk entries, far tiles), the walk queue, the far-tile queue, the far tiles' open lanes
    const size_t far_off = (size_t)WALK_SHARDS * 16 + (size_t)WALK_SHARDS * tiles_per_shard * 64;
    const size_t mask_off = (far_off + (size_t)WALK_SHARDS * tiles_per_shard + 1) / 2 * 2;
    unsigned* qcount = ctx->walk_queue.as<unsigned>();
    unsigned* queue = qcount + WALK_SHARDS * 16;
    // k_icp_ball runs one wave per tile, one tile after the other: with few tiles (a scan of some ten thousand points) that is
    // a handful of long serial jobs, and eight lanes per query in k_icp_walk, spread over the whole GPU, finish sooner
    // (bunny pair, 626 tiles, 30 passes: 12.9 ms with the ball kernel, 5 ms without)
    const bool use_ball = tiles >= ctx->icp_ball_min_tiles;
    unsigned* far_queue = use_ball ? qcount + far_off : nullptr;
    unsigned long long* far_mask = reinterpret_cast<unsigned long long*>(qcount + mask_off);
    // every wave of a group adds once more than it gets tiles, and the group's tiles are added for once each
    const unsigned long long counter_base = ctx->tile_counter_base;
    ctx->tile_counter_base += tiles_per_group + (unsigned long long)(blocks / 8) * TILE_WAVES;
    Mat34 M;
    for (int k = 0; k < 12; k++) M.m[k] = T ? T[k] : 0.0;
    const GridView g = view_of(ctx);
    double* part = ctx->partials.as<double>();
    // A small source cloud (a scan of some ten thousand points) is walked as a whole, eight lanes per point from its previous
    // pair: its few hundred tiles would be a short kernel of long serial jobs followed by a second one for the leftovers
    // (bunny pair, converged pass: 102 us as tile + walk, 65 us walked)
    const bool walk_all = tiles < ctx->icp_walk_all_tiles;
    if (!walk_all)
    hipLaunchKernelGGL(k_icp_tile, dim3(blocks), dim3(64 * TILE_WAVES), 0, ctx->stream, g, ctx->src_pts.as<float4>(), ctx->ns, M, Tdev, ctx->nt, idx_dev,
                       ctx->nn_pair.as<float4>(), tiles_per_group, ctx->tile_counters.as<unsigned long long>(), counter_base, queue, qcount, tiles_per_shard,
                       far_queue, far_mask);
    if (use_ball)
        hipLaunchKernelGGL(k_icp_ball, dim3(blocks), dim3(64 * TILE_WAVES), 0, ctx->stream, g, ctx->src_pts.as<float4>(), ctx->ns, M, Tdev, ctx->nt, idx_dev,
                           ctx->nn_pair.as<float4>(), far_queue, far_mask, queue, qcount, tiles_per_shard, ctx->icp_ball_many, ctx->icp_ball_few);
    if (!walk_all)
        hipLaunchKernelGGL(k_icp_solo, dim3((unsigned)std::min<size_t>((ctx->ns + 255) / 256, (size_t)ctx->num_cus * 8)), dim3(256), 0, ctx->stream, g, ctx->src_pts.as<float4>(), M, Tdev, ctx->nt,
                           queue, qcount, tiles_per_shard, idx_dev, ctx->nn_pair.as<float4>());
    // (what is left to the walk when the tile kernels ran: queries without any candidate, or absurdly far from theirs; the
    // grid is small because launching two thousand workgroups that find nothing to do costs 25 us a pass. A cloud of
    // many millions of points has smaller cells, the same offset is more cells, and the first passes leave the walk
    // hundreds of thousands of queries more than 40 cells from their candidates: there the full grid pays,
    // 16 M pair, first pass: walk kernel 10.9 ms with 512 workgroups)
    const size_t walk_cap = (size_t)ctx->num_cus * (tiles >= ctx->icp_walk_wide_tiles ? 8 : 2);
    const unsigned walk_blocks = (unsigned)(std::min<size_t>((tiles * 2 + WALK_SHARDS - 1) / WALK_SHARDS, walk_cap / WALK_SHARDS) * WALK_SHARDS);
    const unsigned all_blocks = (unsigned)std::min<size_t>((ctx->ns + NN_OCTETS - 1) / NN_OCTETS, (size_t)ctx->num_cus * 8);
    hipLaunchKernelGGL(k_icp_walk, dim3(walk_all ? all_blocks : std::max(walk_blocks, (unsigned)WALK_SHARDS)), dim3(256), 0, ctx->stream, g, ctx->src_pts.as<float4>(), M, Tdev, ctx->nt,
                       queue, qcount, tiles_per_shard, idx_dev, ctx->nn_pair.as<float4>(), walk_all ? ctx->ns : (size_t)0);
    hipLaunchKernelGGL(k_icp_sums, dim3(sblocks), dim3(64 * TILE_WAVES), 0, ctx->stream, g, ctx->src_pts.as<float4>(), ctx->ns, M, Tdev, ctx->nt,
                       ctx->nn_pair.as<float4>(), tiles_per_sblock, part);
    hipLaunchKernelGGL(k_reduce_sums, dim3(ICPX_NSUMS), dim3(256), 0, ctx->stream, part, swaves, sums_dev, qcount, WALK_SHARDS);
#ifdef ICPX_NN_STATS
    {
        unsigned long long h[16];
        hipStreamSynchronize(ctx->stream);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_nn_stats), sizeof h);
        fprintf(stderr, "tiles %llu staged %llu cols/tile %.1f pts/staged tile %.1f; far tiles %llu covered %llu pts/covered %.0f lanes left to the walk %llu\n", h[0], h[1], (double)h[6] / h[0], (double)h[7] / (h[1] ? h[1] : 1),
                h[2], h[3], (double)h[4] / (h[3] ? h[3] : 1), h[5]);
        fprintf(stderr, "   BALL left: near %llu, part<8 %llu, spread %llu, columns %llu, points %llu\n", h[12], h[8], h[9], h[10], h[11]);
        fprintf(stderr, "   WALKQ %llu; tile kernel: ambiguous lanes %llu, lanes of unstaged tiles %llu\n", h[13], h[14], h[15]);
        { unsigned long long bt[8], zz[8] = {0}; hipMemcpyFromSymbol(bt, HIP_SYMBOL(g_ball_t), sizeof bt); hipMemcpyToSymbol(HIP_SYMBOL(g_ball_t), zz, sizeof zz);
          fprintf(stderr, "   BALLT us per far tile: candidates %.1f geometry %.1f columns %.1f staging %.1f compare %.1f resolve %.1f | tile %.1f\n", bt[0] / 100.0 / (h[2] ? h[2] : 1), bt[1] / 100.0 / (h[2] ? h[2] : 1),
                  bt[2] / 100.0 / (h[2] ? h[2] : 1), bt[3] / 100.0 / (h[2] ? h[2] : 1), bt[4] / 100.0 / (h[2] ? h[2] : 1), bt[5] / 100.0 / (h[2] ? h[2] : 1), bt[6] / 100.0 / (h[2] ? h[2] : 1)); }
        {
            static std::vector<unsigned long long> b0(8192), b1(8192), bb(8192), bn(8192);
            hipMemcpyFromSymbol(b0.data(), HIP_SYMBOL(g_ball_t0), 8192 * 8); hipMemcpyFromSymbol(b1.data(), HIP_SYMBOL(g_ball_t1), 8192 * 8);
            hipMemcpyFromSymbol(bb.data(), HIP_SYMBOL(g_ball_busy), 8192 * 8); hipMemcpyFromSymbol(bn.data(), HIP_SYMBOL(g_ball_n), 8192 * 8);
            unsigned long long lo = ~0ull, hi = 0; double busy = 0, mxb = 0; unsigned long long ntl = 0, worked = 0, mxn = 0;
            const int nw = std::min(blocks * TILE_WAVES, 8192);
            for (int b = 0; b < nw; b++) { lo = std::min(lo, b0[b]); if (bn[b]) { hi = std::max(hi, b1[b]); worked++; } busy += (double)bb[b]; mxb = std::max(mxb, (double)bb[b]); ntl += bn[b]; mxn = std::max(mxn, bn[b]); }
            std::vector<double> ends; for (int b = 0; b < nw; b++) if (bn[b]) ends.push_back((b1[b] - lo) / 100.0);
            std::sort(ends.begin(), ends.end());
            fprintf(stderr, "   BALLW waves %d (%llu got tiles, %llu tiles, most per wave %llu): span %.1f us, busy per wave mean %.1f max %.1f us; wave end times p10 %.1f p50 %.1f p90 %.1f max %.1f\n", nw, worked, ntl, mxn,
                    hi > lo ? (hi - lo) / 100.0 : 0.0, busy / nw / 100.0, mxb / 100.0, ends.empty() ? 0.0 : ends[ends.size() / 10], ends.empty() ? 0.0 : ends[ends.size() / 2], ends.empty() ? 0.0 : ends[ends.size() * 9 / 10], ends.empty() ? 0.0 : ends.back());
        }
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_nn_stats), z, sizeof z);
        static std::vector<unsigned long long> t0(8192), t1(8192);
        hipMemcpyFromSymbol(t0.data(), HIP_SYMBOL(g_nn_t0), 8192 * 8);
        hipMemcpyFromSymbol(t1.data(), HIP_SYMBOL(g_nn_t1), 8192 * 8);
        unsigned long long lo = ~0ull, hi = 0, latest_start = 0; double sum = 0, mx = 0;
        for (int b = 0; b < blocks * TILE_WAVES && b < 8192; b++) { lo = std::min(lo, t0[b]); hi = std::max(hi, t1[b]); latest_start = std::max(latest_start, t0[b]); sum += (double)(t1[b] - t0[b]); mx = std::max(mx, (double)(t1[b] - t0[b])); }
        fprintf(stderr, "   waves %d: span %.1f us, mean wave %.1f us, longest %.1f us, last start at %.1f us\n", blocks * TILE_WAVES, (hi - lo) / 100.0, sum / (blocks * TILE_WAVES) / 100.0, mx / 100.0, (latest_start - lo) / 100.0);
        const size_t nt_ = std::min<size_t>(tiles, 1u << 18);
        static std::vector<unsigned> tt(1 << 18), ts(1 << 18), tp(1 << 18);
        hipMemcpyFromSymbol(tt.data(), HIP_SYMBOL(g_nn_tile_t), nt_ * 4); hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_nn_tile_steps), nt_ * 4); hipMemcpyFromSymbol(tp.data(), HIP_SYMBOL(g_nn_tile_total), nt_ * 4);
        std::vector<size_t> ord(nt_); for (size_t k = 0; k < nt_; k++) ord[k] = k;
        std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return tt[a] > tt[b]; });
        double tsum = 0, ssum = 0; for (size_t k = 0; k < nt_; k++) { tsum += tt[k]; ssum += ts[k]; }
        fprintf(stderr, "   tile time: mean %.2f us, median %.2f, p99 %.2f, max %.2f; mean steps %.1f; slowest:", tsum / nt_ / 100.0, tt[ord[nt_ / 2]] / 100.0, tt[ord[nt_ / 100]] / 100.0, tt[ord[0]] / 100.0, ssum / nt_);
        for (int k = 0; k < 5; k++) fprintf(stderr, " [t %.1f us steps %u pts %u]", tt[ord[k]] / 100.0, ts[ord[k]], tp[ord[k]]);
        fprintf(stderr, "\n");
    }
#endif
    return icpx_launch_check(ctx, "k_icp_tile");
}

extern "C" int icpx_icp_step(icpx_ctx* ctx, const double T[16], double sums[ICPX_NSUMS], int32_t* idx_opt, int idx_where) {
    if (!ctx) return ICPX_EINVAL;
    if (!sums) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step: sums is NULL");
    ICPX_TRY(icpx_bind(ctx));
    ICPX_TRY(icpx_reserve(ctx, ctx->sums, ICPX_NSUMS * sizeof(double)));
    int* didx = nullptr;
    if (idx_opt) {
        if (idx_where == ICPX_DEVICE) didx = idx_opt;
        else { ICPX_TRY(icpx_reserve(ctx, ctx->scratch_idx, ctx->ns * sizeof(int))); didx = ctx->scratch_idx.as<int>(); }
    }
    // The reduction kernel writes the 16 sums straight into pinned host memory: no copy command between the
    // kernels and the wait, which matters when a pass over a small cloud takes 0.1 ms.
    if (!ctx->pin_sums) ICPX_CHECK_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->pin_sums), ICPX_NSUMS * sizeof(double), hipHostMallocDefault));
    ICPX_TRY(icp_step_enqueue(ctx, T, ctx->pin_sums, didx));
    if (idx_opt && idx_where != ICPX_DEVICE)
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(idx_opt, didx, ctx->ns * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(sums, ctx->pin_sums, ICPX_NSUMS * sizeof(double));
    return ICPX_OK;
}

extern "C" int icpx_icp_step_trimmed(icpx_ctx* ctx, const double T[16], uint64_t keep, double sums[ICPX_NSUMS]) {
    if (!ctx) return ICPX_EINVAL;
    if (!sums) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step_trimmed: sums is NULL");
    if (!ctx->grid_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_step_trimmed: no target cloud (call icpx_set_target first)");
    if (ctx->ns == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_step_trimmed: no source cloud (call icpx_set_source first)");
    if (!T) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step_trimmed: T is NULL");
    if (keep > ctx->ns) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step_trimmed: keep exceeds the number of source points");
    ICPX_TRY(icpx_bind(ctx));
    const size_t n = ctx->ns;
    ICPX_TRY(icpx_reserve(ctx, ctx->sums, ICPX_NSUMS * sizeof(double)));
    ICPX_TRY(icpx_reserve(ctx, ctx->trim_keys_a, n * 8));
    ICPX_TRY(icpx_reserve(ctx, ctx->trim_keys_b, n * 8));
    ICPX_TRY(icpx_reserve(ctx, ctx->trim_vals_a, n * 4));
    ICPX_TRY(icpx_reserve(ctx, ctx->trim_vals_b, n * 4));
    ICPX_TRY(icpx_reserve(ctx, ctx->trim_p, n * 3 * sizeof(double)));
    ICPX_TRY(icpx_reserve(ctx, ctx->trim_idx, n * sizeof(int)));
    const int blocks = (int)std::min<size_t>((std::max<size_t>(keep, 1) + 255) / 256, (size_t)ctx->num_cus * 32);
    ICPX_TRY(icpx_reserve(ctx, ctx->partials, (size_t)blocks * ICPX_NSUMS * sizeof(double)));
    Mat34 M;
    for (int k = 0; k < 12; k++) M.m[k] = T[k];
    hipLaunchKernelGGL(k_icp_pairs, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, ctx->stream, view_of(ctx), ctx->src_pts.as<float4>(), n, M,
                       ctx->trim_keys_a.as<unsigned long long>(), ctx->trim_vals_a.as<unsigned>(), ctx->trim_p.as<double>(),
                       ctx->trim_idx.as<int>(), ctx->nt, ctx->nn_seed.as<int>());
    ICPX_TRY(icpx_launch_check(ctx, "k_icp_pairs"));
    ICPX_TRY(icpx_sort_pairs64(ctx, n, ctx->trim_keys_a.as<unsigned long long>(), ctx->trim_keys_b.as<unsigned long long>(),
                               ctx->trim_vals_a.as<unsigned>(), ctx->trim_vals_b.as<unsigned>()));
    hipLaunchKernelGGL(k_icp_trim_sums, dim3(blocks), dim3(256), 0, ctx->stream, ctx->trim_vals_b.as<unsigned>(),
                       ctx->trim_keys_b.as<unsigned long long>(), (size_t)keep, ctx->trim_p.as<double>(), ctx->trim_idx.as<int>(),
                       ctx->tgt_xyz.as<float>(), ctx->partials.as<double>());
    hipLaunchKernelGGL(k_reduce_sums, dim3(ICPX_NSUMS), dim3(256), 0, ctx->stream, ctx->partials.as<double>(), blocks, ctx->sums.as<double>());
    ICPX_TRY(icpx_launch_check(ctx, "k_icp_trim_sums"));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(sums, ctx->sums.p, ICPX_NSUMS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

// ---- a whole registration loop without the host in between -----------------------------------------------------------
// SingleThreadIcp::registration (src/SingleThreadIcp.cpp:293-327) runs a fixed number of iterations, each of which
// needs the previous one's transform and nothing else from the host: the sixteen sums of a pass stay on the device, one
// thread turns them into the iteration's rigid update with the arithmetic of the host classes (icpx_solve.h: the same
// statements compiled for both sides) and multiplies it onto the accumulated transform, which the kernels of the next
// pass read from device memory.
struct IcpLoopState {
    double fin[16];          // the accumulated transform (finMat), row-major 4 x 4
    double mean_tar[3];      // the reference's running target mean (carried from iteration to iteration, :229-231)
    double pad;
};

__global__ void k_icp_solve(const double* __restrict__ sums, IcpLoopState* __restrict__ st, double N, double denom, double* __restrict__ trace) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[ICPX_NSUMS], mean_tar[3], mean_src[3], H[9], T[16], fin[16], fin_old[16], mse_after;
    for (int k = 0; k < ICPX_NSUMS; k++) s[k] = sums[k];
    for (int k = 0; k < 3; k++) mean_tar[k] = st->mean_tar[k];
    for (int k = 0; k < 16; k++) fin_old[k] = st->fin[k];
    icpx_solve::covariance(s, N, denom, mean_tar, mean_src, H);
    icpx_solve::best_fit(H, mean_tar, mean_src, s, N, T, &mse_after);
    icpx_solve::mul4(T, fin_old, fin);
    for (int k = 0; k < 16; k++) st->fin[k] = fin[k];
    for (int k = 0; k < 3; k++) st->mean_tar[k] = mean_tar[k];
    if (trace) {
        for (int k = 0; k < 16; k++) trace[k] = T[k];
        trace[16] = mse_after;
        trace[17] = s[15] / N;
    }
}

extern "C" int icpx_icp_run(icpx_ctx* ctx, const double T0[16], int iterations, double mean_tar[3], double T_out[16], double* trace) {
    if (!ctx) return ICPX_EINVAL;
    if (!T0 || !mean_tar || !T_out) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_run: NULL argument");
    if (iterations < 0 || iterations > 1000000) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_run: iterations must be 0 .. 1000000");
    if (!ctx->grid_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_run: no target cloud (call icpx_set_target first)");
    if (ctx->ns == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_run: no source cloud (call icpx_set_source first)");
    ICPX_TRY(icpx_bind(ctx));
    const size_t state_doubles = sizeof(IcpLoopState) / sizeof(double), trace_doubles = 18 * (size_t)iterations;
    ICPX_TRY(icpx_reserve(ctx, ctx->icp_loop, (state_doubles + ICPX_NSUMS + trace_doubles) * sizeof(double)));
    double* base = ctx->icp_loop.as<double>();
    IcpLoopState* st = reinterpret_cast<IcpLoopState*>(base);
    double* sums_dev = base + state_doubles;
    double* trace_dev = sums_dev + ICPX_NSUMS;
    IcpLoopState h;
    for (int k = 0; k < 16; k++) h.fin[k] = T0[k];
    for (int k = 0; k < 3; k++) h.mean_tar[k] = mean_tar[k];
    h.pad = 0;
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(st, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (h is on this call's stack)
    const double N = (double)ctx->ns, denom = (double)(float)ctx->ns;   // "/= (float)Nd", src/SingleThreadIcp.cpp:244
    for (int it = 0; it < iterations; it++) {
        ICPX_TRY(icp_step_enqueue(ctx, nullptr, sums_dev, nullptr, reinterpret_cast<const Mat34*>(st->fin)));
        hipLaunchKernelGGL(k_icp_solve, dim3(1), dim3(64), 0, ctx->stream, sums_dev, st, N, denom, trace_dev + 18 * (size_t)it);
    }
    ICPX_TRY(icpx_launch_check(ctx, "k_icp_solve"));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    if (trace && iterations) ICPX_CHECK_HIP(ctx, hipMemcpyAsync(trace, trace_dev, trace_doubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 16; k++) T_out[k] = h.fin[k];
    for (int k = 0; k < 3; k++) mean_tar[k] = h.mean_tar[k];
    return ICPX_OK;
}

extern "C" int icpx_icp_step_async(icpx_ctx* ctx, const double T[16], double* sums_dev) {
    if (!ctx) return ICPX_EINVAL;
    if (!sums_dev) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step_async: sums_dev is NULL");
    ICPX_TRY(icpx_bind(ctx));
    return icp_step_enqueue(ctx, T, sums_dev, nullptr);
}
