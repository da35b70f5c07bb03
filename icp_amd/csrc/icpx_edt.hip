// Exact squared Euclidean distance transform on a div^3 voxel grid, and its lookup.
//
// Stands in for LDT::LDT and LDT::Distance of the reference (src/testLDT.cpp:40-212, :235-290).
// The result is the grid the reference's three passes compute before the final sqrt (with the y-pass
// scan-back comparing y against t[q], as its x pass does at :197-202), i.e. the exact squared EDT of
// the occupied voxels in cell units.
//
// Pipeline (all on one stream):
//   splat    points -> occupancy bits (one bit per voxel, 64-bit words along z)      src/testLDT.cpp:74-88
//   census   per plane x: records of the rows y that hold any occupied voxel; list of non-empty planes
//   pass zy  per plane x and 64-wide z chunk (one lane per z): distance along z from the bit row
//            (:91-109), then the lower envelope along y (:123-157); rows without voxels never win the
//            envelope, so only listed rows are scanned
//   pass x   per y and z chunk: lower envelope along x over the non-empty planes, writes the final
//            squared grid (:172-206 before the sqrt)
// Memory traffic: the only full-size arrays are the in-plane distances (written for non-empty planes,
// read once) and the output; the occupancy is 1 bit per voxel and stays in L2.
//
// Envelope stacks: one lane owns one grid row; its stack of (site, start, value) entries lives in LDS
// for the first `cap` levels ([level][lane], conflict-free) and, beyond that, in the lane's own
// not-yet-written output slots (slot `level` of the row), which the scan-back has not reached when it
// reads them back (level <= start[level] <= current position).
#include "icpx_ctx.h"
#include "icpx_edt_dev.h"

#include <algorithm>
#include <cstdlib>

typedef unsigned long long u64;

namespace {

constexpr int BIGD = 1 << 20;

// ---- splat ------------------------------------------------------------------------------------------
__global__ void k_edt_splat(const float* __restrict__ pts, size_t n, float mnx, float mny, float mnz, float cell, int div,
                            int W, u64* __restrict__ bits, u64* __restrict__ counters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // float subtract, float divide (round to nearest), truncate: src/testLDT.cpp:78-80
    int ix = (int)__fdiv_rn(__fsub_rn(pts[3 * i], mnx), cell);
    int iy = (int)__fdiv_rn(__fsub_rn(pts[3 * i + 1], mny), cell);
    int iz = (int)__fdiv_rn(__fsub_rn(pts[3 * i + 2], mnz), cell);
    if (ix < 0 || iy < 0 || iz < 0 || ix >= div || iy >= div || iz >= div) {
        atomicAdd(&counters[1], 1ull);
        return;
    }
    atomicOr(&bits[((size_t)ix * div + iy) * W + (iz >> 6)], 1ull << (iz & 63));
}

// ---- census: one wave per plane x ------------------------------------------------------------------------
// For every non-empty row (x, y) and every 64-wide z chunk c the census writes one 16-byte record that
// pass zy consumes: the chunk's own occupancy word, the nearest set bit below / above the chunk, and y.
// Records of one (x, c) are contiguous in the order of ascending y, so pass zy streams them.
struct RowRec {
    u64 own;                // occupancy bits of z in [64c, 64c + 63]
    unsigned short below;   // 1 + position of the nearest set bit in lower words, 0 = none
    unsigned short above;   // position of the nearest set bit in higher words, 0xffff = none
    int y;                  // row index (the envelope site)
};

__global__ __launch_bounds__(64) void k_edt_census(const u64* __restrict__ bits, int div, int W, RowRec* __restrict__ recs,
                                                   int* __restrict__ rowcnt, u64* __restrict__ counters) {
    const int x = blockIdx.x, lane = threadIdx.x;
    int cnt = 0;
    unsigned occ = 0;
    for (int base = 0; base < div; base += 64) {
        const int y = base + lane;
        bool any = false;
        const u64* row = bits + ((size_t)x * div + (y < div ? y : 0)) * W;
        if (y < div)
            for (int w = 0; w < W; w++) { u64 v = row[w]; any |= v != 0; occ += __popcll(v); }
        const u64 m = __ballot(any);
        if (any) {
            const int k = cnt + __popcll(m & ((1ull << lane) - 1));
            int below = 0;   // running: 1 + highest set position seen in words < c
            for (int c = 0; c < W; c++) {
                int above = 0xffff;
                for (int j = c + 1; j < W; j++) { u64 v = row[j]; if (v) { above = 64 * j + (__ffsll((long long)v) - 1); break; } }
                RowRec r;
                r.own = row[c]; r.below = (unsigned short)below; r.above = (unsigned short)above; r.y = y;
                recs[((size_t)x * W + c) * div + k] = r;
                if (r.own) below = 64 * c + 64 - __clzll(r.own);
            }
        }
        cnt += __popcll(m);
    }
    for (int o = 32; o > 0; o >>= 1) occ += __shfl_xor(occ, o);
    if (lane == 0) {
        rowcnt[x] = cnt;
        if (occ) atomicAdd(&counters[0], (u64)occ);
    }
}

// Ascending list of the non-empty planes as int32, padded with copies of the last entry up to
// `padded` entries so that pass x can prefetch whole groups without bounds checks.
__global__ __launch_bounds__(64) void k_edt_planes(const int* __restrict__ rowcnt, int div, int padded, int* __restrict__ planelist,
                                                   u64* __restrict__ counters) {
    const int lane = threadIdx.x;
    int cnt = 0, last = 0;
    for (int base = 0; base < div; base += 64) {
        const int x = base + lane;
        const bool any = x < div && rowcnt[x] > 0;
        const u64 m = __ballot(any);
        if (any) planelist[cnt + __popcll(m & ((1ull << lane) - 1))] = x;
        if (m) last = base + 63 - __clzll(m);
        cnt += __popcll(m);
    }
    for (int k = cnt + lane; k < padded; k += 64) planelist[k] = last;
    if (lane == 0) counters[2] = (u64)cnt;
}

// floor(N / D) for 0 <= N < 2^23, 1 <= D < 2^12 via a float reciprocal and correction steps.
__device__ __forceinline__ int floordiv_pos(int N, int D) {
    int k = (int)((float)N * __builtin_amdgcn_rcpf((float)D));
    int r = N - k * D;
    if (r < 0) { k--; r += D; }
    if (r < 0) { k--; r += D; }
    if (r >= D) { k++; r -= D; }
    if (r >= D) k++;
    return k;
}

__device__ __forceinline__ int bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---- pass zy ------------------------------------------------------------------------------------------
// One wave = plane x, z chunk c; lane = z. The records of the plane's non-empty rows are fetched 64 at a
// time (lane j holds record j, one coalesced 1 KiB load, the next block prefetched) and broadcast with
// v_readlane as the scan reaches them, so the serial scan never waits on memory.
// Stack entry: site (10 bits) | start (10 bits) | distance along z (10 bits).
// `plane` and `spill_base` are the same array: results are stored through the first, spilled stack
// entries are written and read through the second. Declaring them as unrelated pointers lets the
// scan-back keep its result stores in flight; the only same-address pairs (a spilled entry and the
// result that later overwrites its slot) are ordered by the data dependency on the popped entry.
__global__ __launch_bounds__(64) void k_edt_pass_zy(const RowRec* __restrict__ recs, const int* __restrict__ rowcnt,
                                                    int* __restrict__ plane, int* __restrict__ spill_base, int div, int W, int cap) {
    extern __shared__ unsigned lds[];   // [cap][64]
    const int x = blockIdx.x / W, c = blockIdx.x % W;
    const int cnt = rowcnt[x];
    if (cnt == 0) return;   // plane without voxels: never read by pass x
    const int lane = threadIdx.x;
    const bool active = c * 64 + lane < div;
    const int z = active ? c * 64 + lane : div - 1;      // idle lanes shadow the last valid z and never store
    const int zl = z - c * 64;                            // bit position inside the chunk word
    const size_t rowstride = (size_t)div;
    int* __restrict__ out = plane + (size_t)x * div * div + z;         // out[y * div]
    int* __restrict__ spill = spill_base + (size_t)x * div * div + z;  // spill[level * div]
    const RowRec* rr = recs + ((size_t)x * W + c) * div;

    int q = -1, ts = 0, tt = 0, tg = 0;   // cached top entry
    const int4* rr4 = reinterpret_cast<const int4*>(rr);
    int4 nxt = lane < cnt ? rr4[lane] : make_int4(0, 0, 0, 0);
    for (int k0 = 0; k0 < cnt; k0 += 64) {
        const int4 cur = nxt;
        if (k0 + 64 + lane < cnt) nxt = rr4[k0 + 64 + lane];
        const int m = min(64, cnt - k0);
        for (int j = 0; j < m; j++) {
            const unsigned lo = (unsigned)bcast(cur.x, j), hi = (unsigned)bcast(cur.y, j);
            const unsigned ba = (unsigned)bcast(cur.z, j);
            const int u = bcast(cur.w, j);
            const u64 own = ((u64)hi << 32) | lo;
            const int below = (int)(ba & 0xffff) - 1, above = (int)(ba >> 16);
            const u64 md = own & (~0ull >> (63 - zl));
            const u64 mu = own >> zl;
            const int down = md ? zl - (63 - __clzll(md)) : (below >= 0 ? z - below : BIGD);
            const int up = mu ? (__ffsll((long long)mu) - 1) : (above != 0xffff ? above - z : BIGD);
            const int g = min(down, up);
            const int Gu = g * g;

            if (q >= 0) {
                while (true) {
                    const int a = (tt - ts) * (tt - ts) + tg * tg;
                    const int b = (tt - u) * (tt - u) + Gu;
                    if (a <= b) break;
                    q--;
                    if (q < 0) break;
                    if (q < cap) { const unsigned e = lds[q * 64 + lane]; ts = e & 1023; tt = (e >> 10) & 1023; tg = e >> 20; }
                    else { const unsigned e = (unsigned)spill[(size_t)q * rowstride]; ts = e & 1023; tt = (e >> 10) & 1023; tg = e >> 20; }
                }
            }
            bool push = false;
            int w = 0;
            if (q < 0) { q = 0; push = true; }
            else {
                w = 1 + floordiv_pos(u * u - ts * ts + Gu - tg * tg, 2 * (u - ts));
                if (w < div) { q++; push = true; }
            }
            if (push) {
                ts = u; tt = w; tg = g;
                const unsigned e = (unsigned)u | ((unsigned)w << 10) | ((unsigned)g << 20);
                if (q < cap) lds[q * 64 + lane] = e;
                else if (active) spill[(size_t)q * rowstride] = (int)e;
            }
        }
    }
    for (int y = div - 1; y >= 0; y--) {
        const int v = (y - ts) * (y - ts) + tg * tg;
        if (active) out[(size_t)y * rowstride] = v;
        if ((y == tt) && q > 0) {
            q--;
            if (q < cap) { const unsigned e = lds[q * 64 + lane]; ts = e & 1023; tt = (e >> 10) & 1023; tg = e >> 20; }
            else { const unsigned e = (unsigned)spill[(size_t)q * rowstride]; ts = e & 1023; tt = (e >> 10) & 1023; tg = e >> 20; }
        }
    }
}

// ---- pass x ---------------------------------------------------------------------------------------------
// One wave = row y, z chunk c; lane = z. Sites are the non-empty planes; their in-plane distances are
// fetched PF planes ahead of the scan. A stack entry is {site, start, value}; value needs up to 22 bits,
// so an entry takes 5 bytes for div <= 511 (9 + 9 + 20 bits + 2 spare: a dword plus a byte) and 6 bytes
// otherwise (10 + 10 + 22: a dword plus a halfword), which lets six waves per CU keep 85 / 71 levels in
// LDS. Deeper levels go to the lane's output slots as {site | start << 16} and re-read the value.
// The top two entries are kept in registers, so a pop costs no LDS round trip unless pops follow at once.
constexpr int EDT_PF = 16;

template <bool SMALL> struct XEntry;
template <> struct XEntry<true> {     // div <= 511
    typedef unsigned char hi_t;
    static __device__ __forceinline__ void pack(int s, int t, int G, unsigned& lo, hi_t& hi) {
        lo = (unsigned)G | ((unsigned)s << 20) | ((unsigned)(t & 7) << 29);
        hi = (hi_t)(t >> 3);
    }
    static __device__ __forceinline__ void unpack(unsigned lo, unsigned hi, int& s, int& t, int& G) {
        G = lo & 0xfffff; s = (lo >> 20) & 511; t = (int)(lo >> 29) | ((int)hi << 3);
    }
};
template <> struct XEntry<false> {    // div <= 1023
    typedef unsigned short hi_t;
    static __device__ __forceinline__ void pack(int s, int t, int G, unsigned& lo, hi_t& hi) {
        lo = (unsigned)G | ((unsigned)s << 22);
        hi = (hi_t)t;
    }
    static __device__ __forceinline__ void unpack(unsigned lo, unsigned hi, int& s, int& t, int& G) {
        G = lo & 0x3fffff; s = lo >> 22; t = (int)hi;
    }
};

template <bool SMALL, int ABL = 0>
__global__ __launch_bounds__(64) void k_edt_pass_x(const int* __restrict__ plane, const int* __restrict__ planelist,
                                                   const u64* __restrict__ counters, int* __restrict__ sq, int* __restrict__ spill_base,
                                                   int div, int nchunk, int cap) {
    typedef XEntry<SMALL> E;
    typedef typename E::hi_t hi_t;
    extern __shared__ unsigned lds[];   // [cap][64] dwords, then [cap][64] hi_t
    hi_t* lds_h = reinterpret_cast<hi_t*>(lds + cap * 64);
    const int y = blockIdx.x / nchunk, c = blockIdx.x % nchunk;
    const int lane = threadIdx.x;
    const bool active = c * 64 + lane < div;
    const int z = active ? c * 64 + lane : div - 1;
    int np = (int)counters[2];
    if (ABL == 3) np = min(np, 16);
    const size_t xs = (size_t)div * div;                  // elements between consecutive x
    const int* in = plane + (size_t)y * div + z;          // in[u * xs]
    int* __restrict__ out = sq + (size_t)y * div + z;                  // out[x * xs]
    int* __restrict__ spill = spill_base + (size_t)y * div + z;        // same array as `sq` (see pass zy)

    // loads entry `lvl` of this lane's stack
    auto fetch = [&](int lvl, int& s_, int& t_, int& G_) {
        if (lvl < cap) E::unpack(lds[lvl * 64 + lane], lds_h[lvl * 64 + lane], s_, t_, G_);
        else { const unsigned e = (unsigned)spill[(size_t)lvl * xs]; s_ = e & 0xffff; t_ = e >> 16; G_ = in[(size_t)s_ * xs]; }
    };

    int q = -1;                       // index of the top entry
    int ts = 0, tt = 0, tG = 0;       // top entry
    int bs = 0, bt = 0, bG = 0;       // entry below the top (valid when q >= 1)
    // planelist is padded (see k_edt_planes), so the group loads below need no bounds checks; the
    // indices are wave-uniform scalar loads, the values one coalesced 256-byte load per plane.
    int Gn[EDT_PF], Un[EDT_PF];
#pragma unroll
    for (int j = 0; j < EDT_PF; j++) Un[j] = planelist[j];
#pragma unroll
    for (int j = 0; j < EDT_PF; j++) Gn[j] = in[(size_t)Un[j] * xs];
    for (int k0 = 0; k0 < np; k0 += EDT_PF) {
        int Gc[EDT_PF], Uc[EDT_PF];
#pragma unroll
        for (int j = 0; j < EDT_PF; j++) { Gc[j] = Gn[j]; Uc[j] = Un[j]; }
#pragma unroll
        for (int j = 0; j < EDT_PF; j++) Un[j] = planelist[k0 + EDT_PF + j];
#pragma unroll
        for (int j = 0; j < EDT_PF; j++) Gn[j] = in[(size_t)Un[j] * xs];
#pragma unroll
        for (int j = 0; j < EDT_PF; j++) {
            if (k0 + j < np) {
                const int u = Uc[j], Gu = Gc[j];
                while (q >= 0) {
                    const int a = (tt - ts) * (tt - ts) + tG;
                    const int b = (tt - u) * (tt - u) + Gu;
                    if (a <= b) break;
                    q--;                                   // pop: the entry below becomes the top
                    ts = bs; tt = bt; tG = bG;
                    if (q >= 1) fetch(q - 1, bs, bt, bG);
                }
                bool push = false;
                int w = 0;
                if (q < 0) { q = 0; push = true; }
                else {
                    w = 1 + floordiv_pos(u * u - ts * ts + Gu - tG, 2 * (u - ts));
                    if (w < div) { q++; push = true; }
                }
                if (push) {
                    bs = ts; bt = tt; bG = tG;
                    ts = u; tt = w; tG = Gu;
                    if (q < cap) {
                        unsigned lo; hi_t hi;
                        E::pack(u, w, Gu, lo, hi);
                        lds[q * 64 + lane] = lo; lds_h[q * 64 + lane] = hi;
                    } else if (active) spill[(size_t)q * xs] = (int)((unsigned)u | ((unsigned)w << 16));
                }
            }
        }
    }
    int chk = 0;
    for (int x = (ABL == 2 ? 0 : div - 1); x >= 0; x--) {
        const int v = (x - ts) * (x - ts) + tG;
        if (ABL == 1) chk += v;
        else if (active) out[(size_t)x * xs] = v;
        if ((x == tt) && q > 0) {
            q--;
            ts = bs; tt = bt; tG = bG;
            if (q >= 1) fetch(q - 1, bs, bt, bG);
        }
    }
    if (ABL == 1 && active) out[0] = chk;
}

// ---- lookup (device function in icpx_edt_dev.h) ----
__global__ void k_edt_lookup(const int* __restrict__ sq, int div, float mnx, float mny, float mnz, float cell,
                             const float* __restrict__ q, size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = edt_distance(sq, div, mnx, mny, mnz, cell, q[3 * i], q[3 * i + 1], q[3 * i + 2]);
}

}  // namespace

// ev: NULL, or ICPX_EDT_NSTAGES + 1 events recorded around the stages (clear, splat, census, pass zy, pass x).
static int edt_run_impl(icpx_ctx* ctx, hipEvent_t* ev) {
    const int div = (int)ctx->div, W = ctx->edt_words;
    const int nchunk = (div + 63) / 64;
    hipStream_t s = ctx->stream;
    if (ev) ICPX_CHECK_HIP(ctx, hipEventRecord(ev[0], s));
    ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->edt_bits.p, 0, (size_t)div * div * W * sizeof(u64), s));
    ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->edt_counters.p, 0, 4 * sizeof(u64), s));
    if (ev) ICPX_CHECK_HIP(ctx, hipEventRecord(ev[1], s));
    hipLaunchKernelGGL(k_edt_splat, dim3((unsigned)((ctx->nt + 255) / 256)), dim3(256), 0, s, ctx->tgt_xyz.as<float>(), ctx->nt,
                       ctx->edt_min[0], ctx->edt_min[1], ctx->edt_min[2], ctx->edt_cell, div, W, ctx->edt_bits.as<u64>(),
                       ctx->edt_counters.as<u64>());
    if (ev) ICPX_CHECK_HIP(ctx, hipEventRecord(ev[2], s));
    hipLaunchKernelGGL(k_edt_census, dim3(div), dim3(64), 0, s, ctx->edt_bits.as<u64>(), div, W, ctx->edt_rowlist.as<RowRec>(),
                       ctx->edt_rowcnt.as<int>(), ctx->edt_counters.as<u64>());
    hipLaunchKernelGGL(k_edt_planes, dim3(1), dim3(64), 0, s, ctx->edt_rowcnt.as<int>(), div, (div + 15) / 16 * 16 + 16,
                       ctx->edt_planelist.as<int>(), ctx->edt_counters.as<u64>());
    if (ev) ICPX_CHECK_HIP(ctx, hipEventRecord(ev[3], s));
    hipLaunchKernelGGL(k_edt_pass_zy, dim3(div * nchunk), dim3(64), (size_t)ctx->edt_cap_zy * 64 * 4, s, ctx->edt_rowlist.as<RowRec>(),
                       ctx->edt_rowcnt.as<int>(), ctx->edt_plane.as<int>(), ctx->edt_plane.as<int>(), div, W, ctx->edt_cap_zy);
    if (ev) ICPX_CHECK_HIP(ctx, hipEventRecord(ev[4], s));
    static const int abl = getenv("ICPX_EDT_ABLATE") ? atoi(getenv("ICPX_EDT_ABLATE")) : 0;
    auto kx = abl == 1 ? k_edt_pass_x<true, 1> : abl == 2 ? k_edt_pass_x<true, 2> : abl == 3 ? k_edt_pass_x<true, 3> : k_edt_pass_x<true, 0>;
    if (div <= 511)
        hipLaunchKernelGGL(kx, dim3(div * nchunk), dim3(64), (size_t)ctx->edt_cap_x * 64 * 5, s, ctx->edt_plane.as<int>(),
                           ctx->edt_planelist.as<int>(), ctx->edt_counters.as<u64>(), ctx->edt_sq.as<int>(), ctx->edt_sq.as<int>(),
                           div, nchunk, ctx->edt_cap_x);
    else
        hipLaunchKernelGGL(k_edt_pass_x<false>, dim3(div * nchunk), dim3(64), (size_t)std::min(ctx->edt_cap_x, 71) * 64 * 6, s,
                           ctx->edt_plane.as<int>(), ctx->edt_planelist.as<int>(), ctx->edt_counters.as<u64>(),
                           ctx->edt_sq.as<int>(), ctx->edt_sq.as<int>(), div, nchunk, std::min(ctx->edt_cap_x, 71));
    if (ev) ICPX_CHECK_HIP(ctx, hipEventRecord(ev[5], s));
    return icpx_launch_check(ctx, "EDT kernels");
}

int icpx_edt_run(icpx_ctx* ctx) { return edt_run_impl(ctx, nullptr); }

extern "C" int icpx_edt_rebuild_timed(icpx_ctx* ctx, int reps, float stage_ms[ICPX_EDT_NSTAGES]) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_rebuild_timed: no EDT has been built yet");
    if (reps < 1 || !stage_ms) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_rebuild_timed: bad arguments");
    ICPX_TRY(icpx_bind(ctx));
    std::vector<hipEvent_t> ev((size_t)reps * (ICPX_EDT_NSTAGES + 1));
    for (auto& e : ev) ICPX_CHECK_HIP(ctx, hipEventCreate(&e));
    int rc = ICPX_OK;
    for (int r = 0; r < reps && rc == ICPX_OK; r++) rc = edt_run_impl(ctx, ev.data() + (size_t)r * (ICPX_EDT_NSTAGES + 1));
    if (rc == ICPX_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e)); rc = ICPX_EHIP; }
    }
    for (int k = 0; k < ICPX_EDT_NSTAGES; k++) stage_ms[k] = 0.f;
    for (int r = 0; r < reps && rc == ICPX_OK; r++)
        for (int k = 0; k < ICPX_EDT_NSTAGES; k++) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[(size_t)r * (ICPX_EDT_NSTAGES + 1) + k], ev[(size_t)r * (ICPX_EDT_NSTAGES + 1) + k + 1]);
            stage_ms[k] += ms;
        }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

static int edt_prepare(icpx_ctx* ctx, uint32_t div) {
    if (ctx->nt == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "EDT build: no target cloud (call icpx_set_target first)");
    if (div < 2 || div > ICPX_MAX_DIV) ICPX_FAIL(ctx, ICPX_ELIMIT, "EDT build: div must be in [2, %d]", ICPX_MAX_DIV);
    ICPX_TRY(icpx_bind(ctx));
    ctx->edt_ready = false;
    ctx->div = div;
    ctx->edt_words = (int)((div + 63) / 64);
    const size_t n3 = (size_t)div * div * div;
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_bits, (size_t)div * div * ctx->edt_words * sizeof(u64)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_rowlist, (size_t)div * div * ctx->edt_words * 16));   // RowRec records
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_rowcnt, (size_t)div * sizeof(int)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_planelist, (size_t)((div + 15) / 16 * 16 + 16) * sizeof(int)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_counters, 4 * sizeof(u64)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_plane, n3 * sizeof(int)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_sq, n3 * sizeof(int)));
    return ICPX_OK;
}

extern "C" int icpx_edt_build_box(icpx_ctx* ctx, uint32_t div, const float min3[3], float cell_len) {
    if (!ctx) return ICPX_EINVAL;
    if (!min3 || !(cell_len > 0.f)) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_build_box: bad box");
    ICPX_TRY(edt_prepare(ctx, div));
    ctx->edt_min[0] = min3[0]; ctx->edt_min[1] = min3[1]; ctx->edt_min[2] = min3[2];
    ctx->edt_cell = cell_len;
    ICPX_TRY(icpx_edt_run(ctx));
    u64 counters[4];
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(counters, ctx->edt_counters.p, sizeof counters, hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (counters[0] == 0) ICPX_FAIL(ctx, ICPX_EINVAL, "EDT build: no target point falls inside the grid");
    ctx->edt_ready = true;
    return ICPX_OK;
}

extern "C" int icpx_edt_build(icpx_ctx* ctx, uint32_t div) {
    if (!ctx) return ICPX_EINVAL;
    if (ctx->nt == 0) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_build: no target cloud (call icpx_set_target first)");
    if (div < 2) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_edt_build: div must be >= 2");
    // Box rule of src/testLDT.cpp:54-68 with its arithmetic types: float diagonal, largest extent * 1.1
    // in double, float mid-point, expansion factor 2, cell length rounded to float.
    const float* mn = ctx->bbmin; const float* mx = ctx->bbmax;
    const float d0 = mx[0] - mn[0], d1 = mx[1] - mn[1], d2 = mx[2] - mn[2];
    const float m01 = d0 < d1 ? d1 : d0;
    const float m = m01 < d2 ? d2 : m01;
    double fullLen = (double)m * 1.1;
    const double halfLen = fullLen / 2;
    float lo[3];
    for (int c = 0; c < 3; c++) {
        const float mid = (mn[c] + mx[c]) / 2.0f;
        lo[c] = (float)((double)mid - halfLen * 2.0);
    }
    fullLen = 2 * halfLen * 2.0;
    const float cell = (float)(fullLen / (double)(float)div);
    if (!(cell > 0.f)) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_build: degenerate target bounding box");
    return icpx_edt_build_box(ctx, div, lo, cell);
}

extern "C" int icpx_edt_rebuild_async(icpx_ctx* ctx) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_rebuild_async: no EDT has been built yet");
    ICPX_TRY(icpx_bind(ctx));
    return icpx_edt_run(ctx);
}

extern "C" int icpx_edt_info(icpx_ctx* ctx, uint32_t* div, float min3[3], float* cell_len, uint64_t* occupied, uint64_t* dropped) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_info: no EDT has been built");
    ICPX_TRY(icpx_bind(ctx));
    if (div) *div = ctx->div;
    if (min3) { min3[0] = ctx->edt_min[0]; min3[1] = ctx->edt_min[1]; min3[2] = ctx->edt_min[2]; }
    if (cell_len) *cell_len = ctx->edt_cell;
    if (occupied || dropped) {
        u64 counters[4];
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(counters, ctx->edt_counters.p, sizeof counters, hipMemcpyDeviceToHost, ctx->stream));
        ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (occupied) *occupied = counters[0];
        if (dropped) *dropped = counters[1];
    }
    return ICPX_OK;
}

extern "C" int icpx_edt_read(icpx_ctx* ctx, int32_t* sq_out, int where) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_read: no EDT has been built");
    if (!sq_out) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_read: output is NULL");
    ICPX_TRY(icpx_bind(ctx));
    const size_t bytes = (size_t)ctx->div * ctx->div * ctx->div * sizeof(int);
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(sq_out, ctx->edt_sq.p, bytes, where == ICPX_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                       ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

extern "C" int icpx_edt_grid_ptr(icpx_ctx* ctx, const int32_t** sq_dev) {
    if (!ctx || !sq_dev) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_grid_ptr: no EDT has been built");
    *sq_dev = ctx->edt_sq.as<int32_t>();
    return ICPX_OK;
}

extern "C" int icpx_edt_lookup(icpx_ctx* ctx, const float* xyz, size_t n, int where, float* dist) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_lookup: no EDT has been built");
    if (n == 0) return ICPX_OK;
    if (!xyz || !dist) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_lookup: NULL buffer");
    ICPX_TRY(icpx_bind(ctx));
    const float* dq = xyz; float* dout = dist;
    if (where == ICPX_HOST) {
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * 3 * sizeof(float)));
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_d2, n * sizeof(float)));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        dq = ctx->scratch_q.as<float>(); dout = ctx->scratch_d2.as<float>();
    }
    hipLaunchKernelGGL(k_edt_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->edt_sq.as<int>(), (int)ctx->div,
                       ctx->edt_min[0], ctx->edt_min[1], ctx->edt_min[2], ctx->edt_cell, dq, n, dout);
    ICPX_TRY(icpx_launch_check(ctx, "k_edt_lookup"));
    if (where == ICPX_HOST)
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(dist, dout, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}
