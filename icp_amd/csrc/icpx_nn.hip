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
// float4 {x, y, z, bits(index)}; cell_range[cell] = {first, last}. Queries of the ICP step are the
// source cloud in Morton order, so neighbouring lanes walk neighbouring cells.
#include "icpx_ctx.h"

#include <cfloat>
#include <cmath>
#include <cstring>

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
    const int2* __restrict__ range;
    const float4* __restrict__ pts;
    const unsigned char* __restrict__ masks;   // child masks of all levels >= 1, level l at masks + loff[l]
    const unsigned* __restrict__ loff;          // MAXLEV offsets into masks (device memory)
    int nlev;                                   // number of levels including level 0
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

__global__ void k_cell_ranges(const unsigned* __restrict__ keys, size_t n, int2* __restrict__ range, unsigned* __restrict__ distinct) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = keys[i];
    if (i == 0 || keys[i - 1] != k) { range[k].x = (int)i; if (distinct) atomicAdd(distinct, 1u); }
    if (i == n - 1 || keys[i + 1] != k) range[k].y = (int)(i + 1);
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

__global__ void k_pyr_from_cells(const int2* __restrict__ range, int dx, int dy, int dz, unsigned* __restrict__ up, int uy, int uz) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)dx * dy * dz) return;
    const int2 r = range[i];
    if (r.y <= r.x) return;
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
struct Best {               // the single nearest
    double d2;
    int idx;
    __device__ __forceinline__ void init() { d2 = INFINITY; idx = 0x7fffffff; }
    __device__ __forceinline__ double bound() const { return d2; }
    __device__ __forceinline__ void offer(double d, int id) {
        if (d < d2 || (d == d2 && id < idx)) { d2 = d; idx = id; }
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
    __device__ __forceinline__ void offer(double d, int id) {
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

template <class R>
__device__ __forceinline__ void scan_cell(const GridView& g, int cell, double qx, double qy, double qz, R& b) {
    const int2 r = g.range[cell];
    for (int k = r.x; k < r.y; k++) {
        const float4 p = g.pts[k];
        const double dx = qx - (double)p.x, dy = qy - (double)p.y, dz = qz - (double)p.z;
        const double d = (dx * dx + dy * dy) + dz * dz;
        b.offer(d, __float_as_int(p.w));
    }
}

// Distance from coordinate q to the interval [lo, hi], shrunk by `margin`, squared; 0 inside.
__device__ __forceinline__ double axis_gap2(double q, double lo, double hi, double margin) {
    double g = fmax(fmax(lo - q, q - hi) - margin, 0.0);
    return g * g;
}

// Squared distance from q to the box of cell (i,j,k) at a level whose cells have side `side`, shrunk by
// the margin so that it is a valid lower bound for every point filed under that cell.
__device__ __forceinline__ double box_lb2(double qx, double qy, double qz, double ox, double oy, double oz, double side, int i, int j,
                                          int k, double margin) {
    return axis_gap2(qx, ox + i * side, ox + (i + 1) * side, margin) + axis_gap2(qy, oy + j * side, oy + (j + 1) * side, margin) +
           axis_gap2(qz, oz + k * side, oz + (k + 1) * side, margin);
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
        const int id = __shfl_xor(b.idx, o);
        if (d < b.d2 || (d == b.d2 && id < b.idx)) { b.d2 = d; b.idx = id; }
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
    __device__ __forceinline__ void offer(double d, int id) {
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
        head.d2 = b.l.d2[0]; head.idx = b.l.idx[0];
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

// `snode`, `slb`: the walk's stack in LDS, NN_STACK entries per octet, entry k of octet o at [k * NN_OCTETS + o];
// `b`: the result so far (Best or OctK<K>), initialised, the same in all lanes: empty, or holding candidates
// that are real target points, e.g. the correspondence of the previous ICP iteration, whose distance then prunes
// the search from the start; the result does not depend on them (equal distances still go to the lower index
// because cells are only skipped when their bound is strictly larger).
template <class R>
__device__ R nn_search8(const GridView& g, double qx, double qy, double qz, unsigned* snode, double* slb, const unsigned* loff, R b) {
    const int c = (int)(threadIdx.x & 7);
    const int cx = cell_coord((float)qx, g.minx, g.inv, g.dx), cy = cell_coord((float)qy, g.miny, g.inv, g.dy),
              cz = cell_coord((float)qz, g.minz, g.inv, g.dz);
    const double h = (double)g.h;
    const double margin = 1e-3 * h + 1e-6 * (fabs(qx) + fabs(qy) + fabs(qz));
    const double ox = (double)g.minx, oy = (double)g.miny, oz = (double)g.minz;
    // cells c, c + 8, c + 16, c + 24 of the 3x3x3 block
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int cc = c + 8 * r;
        const int i = cx + cc / 9 - 1, j = cy + (cc / 3) % 3 - 1, k = cz + cc % 3 - 1;
        if (cc < 27 && i >= 0 && i < g.dx && j >= 0 && j < g.dy && k >= 0 && k < g.dz &&
            !(box_lb2(qx, qy, qz, ox, oy, oz, h, i, j, k, margin) > b.bound()))
            scan_cell(g, (i * g.dy + j) * g.dz + k, qx, qy, qz, b);
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
    const int oc = (int)(threadIdx.x >> 3);
    int sp = 1;
    snode[oc] = (unsigned)top << 27;
    slb[oc] = 0.0;
    while (sp > 0) {
        sp--;
        const unsigned node = snode[sp * NN_OCTETS + oc];
        if (slb[sp * NN_OCTETS + oc] > b.bound()) continue;
        const int l = (int)(node >> 27), ni = (int)((node >> 18) & 511), nj = (int)((node >> 9) & 511), nk = (int)(node & 511);
        const unsigned m = g.masks[loff[l] + ((size_t)ni * lev_dim(g.dy, l) + nj) * lev_dim(g.dz, l) + nk];
        const int cl = l - 1;
        const double cside = h * (double)(1 << cl);
        const int ci = 2 * ni + (c >> 2), cj = 2 * nj + ((c >> 1) & 1), ck = 2 * nk + (c & 1);
        const bool has = (m >> c) & 1u;
        const double lb = has ? box_lb2(qx, qy, qz, ox, oy, oz, cside, ci, cj, ck, margin) : INFINITY;
        const bool ok = has && !(lb > b.bound());
        if (cl == 0) {
            // the children are cells: every lane scans its own (those of the first block are done already)
            const bool seen = ci >= cx - 1 && ci <= cx + 1 && cj >= cy - 1 && cj <= cy + 1 && ck >= cz - 1 && ck <= cz + 1;
            if (ok && !seen) scan_cell(g, (ci * g.dy + cj) * g.dz + ck, qx, qy, qz, b);
            octet_merge(b);
            continue;
        }
        const unsigned cand = (unsigned)(__ballot(ok) >> (threadIdx.x & 56)) & 0xffu;
        if (cand == 0) continue;
        const int near = (qx >= ox + (2 * ni + 1) * cside ? 4 : 0) | (qy >= oy + (2 * nj + 1) * cside ? 2 : 0) | (qz >= oz + (2 * nk + 1) * cside ? 1 : 0);
        const unsigned vis = pyr_visit_order(cand, near);
        const int jc = c ^ near, t = jc == 3 ? 4 : jc == 4 ? 3 : jc;          // this lane's place in the visiting order
        const int n = __popc(cand);
        if (ok) {
            const int at = sp + n - 1 - __popc(vis & ((1u << t) - 1u));
            snode[at * NN_OCTETS + oc] = ((unsigned)cl << 27) | ((unsigned)ci << 18) | ((unsigned)cj << 9) | (unsigned)ck;
            slb[at * NN_OCTETS + oc] = lb;
        }
        sp += n;
    }
    return b;
}

__device__ __forceinline__ Best best_empty() { Best b; b.init(); return b; }

// The previous correspondence of a source point as the first candidate: (distance to it now, its index);
// empty when there is none (index < 0) or it does not belong to the current target.
__device__ __forceinline__ Best best_seed(const int* __restrict__ seed, size_t i, size_t nt, const float* __restrict__ tgt_xyz, double qx, double qy,
                                          double qz) {
    Best b;
    b.init();
    if (seed) {
        const int j = seed[i];
        if (j >= 0 && (size_t)j < nt) {
            const double dx = qx - (double)tgt_xyz[3 * (size_t)j], dy = qy - (double)tgt_xyz[3 * (size_t)j + 1], dz = qz - (double)tgt_xyz[3 * (size_t)j + 2];
            b.d2 = (dx * dx + dy * dy) + dz * dz;
            b.idx = j;
        }
    }
    return b;
}

// K nearest target points of every query, ascending by (squared distance, index); slots beyond the number
// of target points get index -1 and an infinite distance.
template <int K>
__global__ __launch_bounds__(256) void k_knn(GridView g, const double* __restrict__ q, size_t n, int kout, int* __restrict__ idx,
                                             double* __restrict__ d2) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ double slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;   // eight lanes per query
    if (i >= n) return;
    OctK<K> b0;
    b0.init();
    const OctK<K> b = nn_search8(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], snode, slb, s_loff, b0);
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
    __shared__ double slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;   // eight lanes per query
    if (i >= n) return;
    const Best b = nn_search8(g, q[3 * i], q[3 * i + 1], q[3 * i + 2], snode, slb, s_loff, best_empty());
    if ((threadIdx.x & 7) == 0) {
        if (idx) idx[i] = b.idx;
        if (d2) d2[i] = b.d2;
    }
}

__global__ __launch_bounds__(256) void k_nn_distance(GridView g, const float* __restrict__ q, size_t n, float* __restrict__ out) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ double slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (i >= n) return;
    const Best b = nn_search8(g, (double)q[3 * i], (double)q[3 * i + 1], (double)q[3 * i + 2], snode, slb, s_loff, best_empty());
    if ((threadIdx.x & 7) == 0) out[i] = (float)__dsqrt_rn(b.d2);
}

struct Mat34 { double m[12]; };

// One ICP pass: p = T * [src;1] (multiply then add, k = 0..3 from a zero accumulator), nearest target
// point q, and block-partial sums {p, q, p q^T, |p-q|^2}.
__global__ __launch_bounds__(256) void k_icp_step(GridView g, const float4* __restrict__ src, size_t n, Mat34 T,
                                                  const float* __restrict__ tgt_xyz, size_t nt, double* __restrict__ partials,
                                                  int* __restrict__ idx_out, int* __restrict__ seed) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ double slb[NN_STACK * NN_OCTETS];
    __shared__ unsigned s_loff[MAXLEV];
    if (threadIdx.x < MAXLEV) s_loff[threadIdx.x] = g.loff[threadIdx.x];
    __syncthreads();
    double acc[ICPX_NSUMS];
#pragma unroll
    for (int k = 0; k < ICPX_NSUMS; k++) acc[k] = 0.0;
    const bool first = (threadIdx.x & 7) == 0;   // eight lanes per source point; the first one keeps the sums
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < n; i += ((size_t)gridDim.x * blockDim.x) >> 3) {
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
        const Best b = nn_search8(g, p[0], p[1], p[2], snode, slb, s_loff, best_seed(seed, i, nt, tgt_xyz, p[0], p[1], p[2]));
        if (!first) continue;
        if (seed) seed[i] = b.idx == 0x7fffffff ? -1 : b.idx;
        if (idx_out) idx_out[__float_as_int(s.w)] = b.idx;
        // a NaN in T leaves no candidate: read point 0 then (the sums are NaN anyway) instead of an invalid index
        const size_t j = b.idx == 0x7fffffff ? 0 : (size_t)b.idx;
        const double q0 = (double)tgt_xyz[3 * j], q1 = (double)tgt_xyz[3 * j + 1], q2 = (double)tgt_xyz[3 * j + 2];
        acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
        acc[3] += q0; acc[4] += q1; acc[5] += q2;
        acc[6] += p[0] * q0; acc[7] += p[0] * q1; acc[8] += p[0] * q2;
        acc[9] += p[1] * q0; acc[10] += p[1] * q1; acc[11] += p[1] * q2;
        acc[12] += p[2] * q0; acc[13] += p[2] * q1; acc[14] += p[2] * q2;
        acc[15] += b.d2;
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
        partials[(size_t)blockIdx.x * ICPX_NSUMS + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// Trimmed ICP pass, first half: p = T * [src;1] and its nearest target point for every source point, stored
// under the caller's source index o: the squared distance as a sort key (the bits of a non-negative double
// order like the value), o itself as the sort value, p and the correspondence.
__global__ __launch_bounds__(256) void k_icp_pairs(GridView g, const float4* __restrict__ src, size_t n, Mat34 T,
                                                   unsigned long long* __restrict__ keys, unsigned* __restrict__ vals,
                                                   double* __restrict__ pbuf, int* __restrict__ idx, const float* __restrict__ tgt_xyz, size_t nt,
                                                   int* __restrict__ seed) {
    __shared__ unsigned snode[NN_STACK * NN_OCTETS];
    __shared__ double slb[NN_STACK * NN_OCTETS];
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
    const Best b = nn_search8(g, p[0], p[1], p[2], snode, slb, s_loff, best_seed(seed, i, nt, tgt_xyz, p[0], p[1], p[2]));
    if ((threadIdx.x & 7) != 0) return;
    if (seed) seed[i] = b.idx == 0x7fffffff ? -1 : b.idx;
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
        const size_t j = idx[o] == 0x7fffffff ? 0 : (size_t)idx[o];   // no candidate (NaN transform): see k_icp_step
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
        partials[(size_t)blockIdx.x * ICPX_NSUMS + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// Fixed-order final reduction of the block partials: thread t sums component t%16 of blocks t/16, t/16+64, ...
__global__ __launch_bounds__(1024) void k_reduce_sums(const double* __restrict__ partials, int nblocks, double* __restrict__ sums) {
    __shared__ double sm[1024];
    const int comp = threadIdx.x & 15, lanegrp = threadIdx.x >> 4;
    double v = 0.0;
    for (int b = lanegrp; b < nblocks; b += 64) v += partials[(size_t)b * ICPX_NSUMS + comp];
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if (lanegrp < s) sm[threadIdx.x] += sm[threadIdx.x + s * 16];
        __syncthreads();
    }
    if (threadIdx.x < ICPX_NSUMS) sums[threadIdx.x] = sm[threadIdx.x];
}

GridView view_of(const icpx_ctx* ctx) {
    GridView g;
    g.dx = ctx->gdim[0]; g.dy = ctx->gdim[1]; g.dz = ctx->gdim[2];
    g.minx = ctx->gmin[0]; g.miny = ctx->gmin[1]; g.minz = ctx->gmin[2];
    g.h = ctx->gh; g.inv = ctx->ginv;
    g.range = ctx->cell_range.as<int2>();
    g.pts = ctx->sorted_pts.as<float4>();
    g.masks = ctx->pyr_masks.as<unsigned char>();
    g.loff = ctx->pyr_tab.as<unsigned>();
    g.nlev = ctx->pyr_nlev;
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
    const double max_cells = 3.2e7;
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
        ICPX_TRY(icpx_reserve(ctx, ctx->cell_range, ncells * sizeof(int2)));
        ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->cell_range.p, 0, ncells * sizeof(int2), ctx->stream));
        unsigned* distinct = reinterpret_cast<unsigned*>(ctx->edt_counters.as<unsigned long long>() + 3);
        ICPX_CHECK_HIP(ctx, hipMemsetAsync(distinct, 0, sizeof(unsigned), ctx->stream));
        GridView g = view_of(ctx);
        hipLaunchKernelGGL(k_cell_keys, dim3(blocks), dim3(256), 0, ctx->stream, ctx->scratch_q.as<float4>(), n, g,
                           ctx->sort_keys_a.as<unsigned>(), ctx->sort_vals_a.as<unsigned>());
        ICPX_TRY(icpx_launch_check(ctx, "k_cell_keys"));
        unsigned bits = 1;
        while ((1ull << bits) < ncells) bits++;
        ICPX_TRY(icpx_sort_pairs(ctx, n, bits));
        hipLaunchKernelGGL(k_cell_ranges, dim3(blocks), dim3(256), 0, ctx->stream, ctx->sort_keys_b.as<unsigned>(), n,
                           ctx->cell_range.as<int2>(), distinct);
        ICPX_TRY(icpx_launch_check(ctx, "k_cell_ranges"));
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
                hipLaunchKernelGGL(k_pyr_from_cells, dim3(nb), dim3(256), 0, ctx->stream, ctx->cell_range.as<int2>(), lo[0], lo[1], lo[2], upw, up[1], up[2]);
            else
                hipLaunchKernelGGL(k_pyr_from_masks, dim3(nb), dim3(256), 0, ctx->stream, fl + ctx->pyr_off[l - 1], lo[0], lo[1], lo[2], upw, up[1], up[2]);
        }
        ICPX_TRY(icpx_launch_check(ctx, "pyramid kernels"));
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
    // correspondences of the last pass, by sorted position (-1 = none): the next pass starts its search from them
    ICPX_TRY(icpx_reserve(ctx, ctx->nn_seed, n * sizeof(int)));
    ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->nn_seed.p, 0xff, n * sizeof(int), ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->ns = n;
    return ICPX_OK;
}

static int icp_step_enqueue(icpx_ctx* ctx, const double T[16], double* sums_dev, int* idx_dev) {
    if (!ctx->grid_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_step: no target cloud (call icpx_set_target first)");
    if (ctx->ns == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_icp_step: no source cloud (call icpx_set_source first)");
    if (!T) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step: T is NULL");
    const int blocks = (int)std::min<size_t>((ctx->ns + 31) / 32, (size_t)ctx->num_cus * 32);   // 32 source points per block
    ICPX_TRY(icpx_reserve(ctx, ctx->partials, (size_t)blocks * ICPX_NSUMS * sizeof(double)));
    Mat34 M;
    for (int k = 0; k < 12; k++) M.m[k] = T[k];
    hipLaunchKernelGGL(k_icp_step, dim3(blocks), dim3(256), 0, ctx->stream, view_of(ctx), ctx->src_pts.as<float4>(), ctx->ns, M,
                       ctx->tgt_xyz.as<float>(), ctx->nt, ctx->partials.as<double>(), idx_dev, ctx->nn_seed.as<int>());
    hipLaunchKernelGGL(k_reduce_sums, dim3(1), dim3(1024), 0, ctx->stream, ctx->partials.as<double>(), blocks, sums_dev);
    return icpx_launch_check(ctx, "k_icp_step");
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
                       ctx->trim_idx.as<int>(), ctx->tgt_xyz.as<float>(), ctx->nt, ctx->nn_seed.as<int>());
    ICPX_TRY(icpx_launch_check(ctx, "k_icp_pairs"));
    ICPX_TRY(icpx_sort_pairs64(ctx, n, ctx->trim_keys_a.as<unsigned long long>(), ctx->trim_keys_b.as<unsigned long long>(),
                               ctx->trim_vals_a.as<unsigned>(), ctx->trim_vals_b.as<unsigned>()));
    hipLaunchKernelGGL(k_icp_trim_sums, dim3(blocks), dim3(256), 0, ctx->stream, ctx->trim_vals_b.as<unsigned>(),
                       ctx->trim_keys_b.as<unsigned long long>(), (size_t)keep, ctx->trim_p.as<double>(), ctx->trim_idx.as<int>(),
                       ctx->tgt_xyz.as<float>(), ctx->partials.as<double>());
    hipLaunchKernelGGL(k_reduce_sums, dim3(1), dim3(1024), 0, ctx->stream, ctx->partials.as<double>(), blocks, ctx->sums.as<double>());
    ICPX_TRY(icpx_launch_check(ctx, "k_icp_trim_sums"));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(sums, ctx->sums.p, ICPX_NSUMS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

extern "C" int icpx_icp_step_async(icpx_ctx* ctx, const double T[16], double* sums_dev) {
    if (!ctx) return ICPX_EINVAL;
    if (!sums_dev) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_icp_step_async: sums_dev is NULL");
    ICPX_TRY(icpx_bind(ctx));
    return icp_step_enqueue(ctx, T, sums_dev, nullptr);
}
