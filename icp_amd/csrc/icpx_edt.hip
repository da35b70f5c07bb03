// Exact squared Euclidean distance transform on a div^3 voxel grid, and its lookup.
//
// Stands in for LDT::LDT and LDT::Distance of the reference (src/testLDT.cpp:40-212, :235-290).
// The result is the grid the reference's three passes compute before the final sqrt (with the y-pass
// scan-back comparing y against t[q], as its x pass does at :197-202), i.e. the exact squared EDT of
// the occupied voxels in cell units.
//
// Pipeline (all on one stream):
//   splat    points -> occupancy bits (one bit per voxel, 64-bit words along z)      src/testLDT.cpp:74-88
//   census   per plane x: records of the rows y that hold any occupied voxel (the list of non-empty planes is
//            put together by one extra workgroup of pass zy)
//   pass zy  per plane x and 64-wide z chunk (one lane per z): distance along z from the bit row
//            (:91-109), then the lower envelope along y (:123-157); rows without voxels never win the
//            envelope, so only the plane's non-empty rows are scanned
//   pass x   per y and z chunk: lower envelope along x over the non-empty planes, writes the final
//            squared grid (:172-206 before the sqrt)
// Memory traffic: the only full-size arrays are the in-plane distances (written for non-empty planes,
// read once) and the output; the occupancy is 1 bit per voxel and stays in L2.
// How the envelope scans are organised: see "lower envelope of parabolas" below.
#include <type_traits>
#include <hip/hip_ext.h>

#include "icpx_ctx.h"
#include "icpx_edt_dev.h"

#include <algorithm>
#include <cstdio>
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

// DT3D's voxel rule (src/jly_3ddt.cpp:30, :792-798): int((p - min) * scale + 0.5) per axis in double, points
// outside the grid skipped.
__global__ void k_edt_splat_round(const float* __restrict__ pts, size_t n, double mnx, double mny, double mnz, double scale, int div,
                                  int W, u64* __restrict__ bits, u64* __restrict__ counters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ix = (int)(((double)pts[3 * i] - mnx) * scale + 0.5);
    const int iy = (int)(((double)pts[3 * i + 1] - mny) * scale + 0.5);
    const int iz = (int)(((double)pts[3 * i + 2] - mnz) * scale + 0.5);
    if (ix < 0 || iy < 0 || iz < 0 || ix >= div || iy >= div || iz >= div) {
        atomicAdd(&counters[1], 1ull);
        return;
    }
    atomicOr(&bits[((size_t)ix * div + iy) * W + (iz >> 6)], 1ull << (iz & 63));
}

// ---- census ------------------------------------------------------------------------------------------------
// For every non-empty row (x, y) and every 64-wide z chunk c the census writes one 16-byte record that
// pass zy consumes: the chunk's own occupancy word, the nearest set bit below / above the chunk, and y.
// Records of one (x, c) are contiguous in the order of ascending y, so pass zy streams them.
struct RowRec {
    u64 own;                // occupancy bits of z in [64c, 64c + 63]
    unsigned short below;   // 1 + position of the nearest set bit in lower words, 0 = none
    unsigned short above;   // position of the nearest set bit in higher words, 0xffff = none
    int y;                  // row index (the envelope site)
};

// One workgroup per plane x, one wave per 64 rows y. A lane reads the occupancy words of its row into
// registers and zeroes them in memory, which leaves the bit array clear for the next build (nothing after
// the census reads it).
constexpr int EDT_MAXW = (ICPX_MAX_DIV + 63) / 64;

__global__ __launch_bounds__(1024) void k_edt_census(u64* __restrict__ bits, int div, int W, RowRec* __restrict__ recs,
                                                     int* __restrict__ rowcnt, u64* __restrict__ counters) {
    __shared__ int wave_cnt[16];
    __shared__ unsigned wave_occ[16];
    const int x = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int y = wv * 64 + lane;
    u64* row = bits + ((size_t)x * div + (y < div ? y : 0)) * W;
    u64 wd[EDT_MAXW];
    unsigned occ = 0;
    bool any = false;
#pragma unroll
    for (int w = 0; w < EDT_MAXW; w++) {
        wd[w] = w < W && y < div ? row[w] : 0ull;
        any |= wd[w] != 0;
        occ += __popcll(wd[w]);
    }
    if (any) {
#pragma unroll
        for (int w = 0; w < EDT_MAXW; w++)
            if (wd[w]) row[w] = 0ull;
    }
    const u64 m = __ballot(any);
    for (int o = 32; o > 0; o >>= 1) occ += __shfl_xor(occ, o);
    if (lane == 0) { wave_cnt[wv] = __popcll(m); wave_occ[wv] = occ; }
    __syncthreads();
    int before = 0, cnt = 0;
    unsigned occ_all = 0;
    for (int w = 0; w < nw; w++) { before += w < wv ? wave_cnt[w] : 0; cnt += wave_cnt[w]; occ_all += wave_occ[w]; }
    if (any) {
        const int k = before + __popcll(m & ((1ull << lane) - 1));
        int above[EDT_MAXW];   // position of the nearest set bit in higher words, 0xffff = none
        int nxt = 0xffff;
#pragma unroll
        for (int c = EDT_MAXW - 1; c >= 0; c--) {
            above[c] = nxt;
            if (wd[c]) nxt = 64 * c + (__ffsll((long long)wd[c]) - 1);
        }
        int below = 0;         // running: 1 + highest set position seen in words < c
#pragma unroll
        for (int c = 0; c < EDT_MAXW; c++) {
            if (c < W) {
                RowRec r;
                r.own = wd[c]; r.below = (unsigned short)below; r.above = (unsigned short)above[c]; r.y = y;
                recs[((size_t)x * W + c) * div + k] = r;
                if (wd[c]) below = 64 * c + 64 - __clzll(wd[c]);
            }
        }
    }
    if (threadIdx.x == 0) {
        rowcnt[x] = cnt;
        if (occ_all) atomicAdd(&counters[0], (u64)occ_all);
    }
}

// The ascending list of the non-empty planes as int32, padded with copies of the last entry up to `padded`
// entries so that pass x can prefetch whole groups without bounds checks; also zeroes the counter set of the
// next build. One wave; run by an extra workgroup of pass zy (the census is complete by then and pass x,
// the only reader, starts after it).
__device__ void edt_list_planes(const int* __restrict__ rowcnt, int div, int padded, int* __restrict__ planelist, u64* __restrict__ counters,
                                u64* __restrict__ next_counters, int lane) {
    int np = 0, last = 0;
    for (int base = 0; base < div; base += 64) {
        const int px = base + lane;
        const bool pany = px < div && rowcnt[px] > 0;
        const u64 pm = __ballot(pany);
        if (pany) planelist[np + __popcll(pm & ((1ull << lane) - 1))] = px;
        if (pm) last = base + 63 - __clzll(pm);
        np += __popcll(pm);
    }
    for (int k = np + lane; k < padded; k += 64) planelist[k] = last;
    if (lane == 0) counters[2] = (u64)np;
    if (lane < 8) next_counters[lane] = 0ull;
}

// floor(N / D) for 0 <= N < 2^22 and even 2 <= D <= 2046: exact whenever the quotient is below 1400, at
// least 1023 otherwise (the callers only tell quotients below div <= 1023 apart). (2N + 1) / (2D) has the
// same floor as N / D and lies at least 1 / (2D) away from every integer, which is more than the error of
// the 1-ulp reciprocal and the rounded product for quotients in that range, so truncating the float
// product is enough; tools/check_floordiv.hip runs through all N and D on the GPU.
__device__ __forceinline__ int floordiv_pos(int N, int D) {
    return (int)((float)(2 * N + 1) * __builtin_amdgcn_rcpf((float)(2 * D)));
}

__device__ __forceinline__ int bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }


// Waits for outstanding global loads only. Placed inside the rarely taken spill branches so that their
// loads are retired before control rejoins the hot path; otherwise the join point gets a wait for every
// outstanding global load, including the prefetches of the next sites.
__device__ __forceinline__ void wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// ---- lower envelope of parabolas: 64 grid rows per workgroup, G waves per row group ---------------------
// Both passes compute out[p] = min over sites u of (p - u)^2 + val[u], p = 0 .. div-1, for one grid row per
// lane (Meijster's scan, src/testLDT.cpp:123-157 / :172-206). A serial scan over all sites of a row is one
// long dependent chain and a div^3 grid only has div^2 rows, too few waves to keep the SIMDs busy, so the
// scan is split: wave g of a workgroup builds the envelope stack of the g-th part of the row's sites (the
// same 64 rows in every wave), then, after a barrier, every wave of the workgroup (the G that scanned and,
// in the default shape, G more that waited for them) evaluates its part of the positions against the stacks
// that can win there and takes the minimum. The stacks are never merged.
//
// A stack entry is {site s, a = s^2 + value} in one dword (s << 22 | a; a <= 3 (div - 1)^2 < 2^22 because
// div <= 1023) and, in a halfword, twice the first position t the entry owns among the entries of its own
// stack (doubled because every use is a comparison against or a product with 2 t). Where the levels live:
//   level < cap                 LDS, [group][level][lane] (conflict-free)
//   cap <= level < cap + 32     LDS pool shared by the workgroup: chunks of 8 levels, handed out with an
//                               atomic counter the first time a stack reaches them; the four chunk ids of
//                               a stack are the bytes of one dword (0xff = none)
//   beyond, or pool exhausted   global spill buffer [level - cap][group][row] of 8-byte entries, allocated
//                               only after a build reported through the overflow counter that it is needed
// Most stacks are a handful of levels deep and a few are several times deeper, so a small private part
// plus the shared pool holds what a fixed split of the same LDS could not. A stack never holds more entries
// than its group has sites; when that fits three workgroups' worth of LDS per CU the host drops the pool
// (POOL = false: every level private, a sentinel row below each stack, the site step hand-written).
constexpr int ENV_VB = 22;
constexpr unsigned ENV_VMASK = (1u << ENV_VB) - 1;
constexpr int ENV_CHUNK = 8;        // levels per pool chunk
constexpr int ENV_MAXCH = 4;        // chunks per stack

template <int G, bool POOL>
struct Env {
    char* lds;
    int cap, nch, lane;
    int off_w, off_t, off_pw, off_pt;   // byte offsets: words [G][ROWS][64] u32, starts [G][ROWS][64] u16, pool words, pool starts
    uint2* sp;               // spill buffer (already offset to this lane's row) or NULL
    int sp_levels;           // levels the spill buffer holds
    unsigned sp_stride;      // entries between consecutive levels of one stack (groups x rows of the pass)
    unsigned sp_gstride;     // entries between the stacks of consecutive groups (rows of the pass)
    u64* ovf;                // overflow counter word

    // LDS header: depths [G][64] u16, chunk tables [G][64] u32, pool counter.
    // Every stack has one more row below its level 0: a sentinel entry {site 0, the largest a, start 0}, which is
    // what the hand-written site step sees as the top of an empty stack.
    static constexpr int FIRST = 1;
    __device__ __forceinline__ int rows() const { return cap + FIRST; }
    __device__ __forceinline__ unsigned short& depth(int g) const { return *reinterpret_cast<unsigned short*>(lds + (g * 64 + lane) * 2); }
    __device__ __forceinline__ unsigned& table(int g) const { return *reinterpret_cast<unsigned*>(lds + G * 128 + (g * 64 + lane) * 4); }
    __device__ __forceinline__ unsigned* counter() const { return reinterpret_cast<unsigned*>(lds + G * 384); }
    // winning stack at the last position of every part of the read-back [parts <= 2 G][64] u8 (in the padding behind the counter)
    __device__ __forceinline__ unsigned char& winner(int g) const { return *reinterpret_cast<unsigned char*>(lds + (POOL ? G * 384 + 64 : G * 128) + g * 64 + lane); }
    // byte addresses of level 0 of group g for this lane
    __device__ __forceinline__ int wbase(int g) const { return off_w + (g * rows() + FIRST) * 256 + lane * 4; }
    __device__ __forceinline__ int tbase(int g) const { return off_t + (g * rows() + FIRST) * 128 + lane * 2; }
    __device__ __forceinline__ unsigned& W(int wb, int level) const { return *reinterpret_cast<unsigned*>(lds + wb + (level << 8)); }
    __device__ __forceinline__ unsigned short& T(int tb, int level) const { return *reinterpret_cast<unsigned short*>(lds + tb + (level << 7)); }
    __device__ __forceinline__ size_t spslot(int g, int level) const { return (size_t)(unsigned)(level - cap) * sp_stride + (size_t)g * sp_gstride; }

    // Pool slot of an overflow level, -1 when the level is not in the pool.
    __device__ __forceinline__ int pool_slot(unsigned tab, int level) const {
        const int L = level - cap, ci = L >> 3;
        if (ci >= ENV_MAXCH) return -1;
        const unsigned id = (tab >> (8 * ci)) & 0xffu;
        return id == 0xffu ? -1 : (int)id * ENV_CHUNK + (L & 7);
    }
    __device__ void load_high(int g, unsigned tab, int level, unsigned& w, int& t) const {
        const int ps = pool_slot(tab, level);
        if (ps >= 0) {
            w = *reinterpret_cast<unsigned*>(lds + off_pw + ps * 4);
            t = *reinterpret_cast<unsigned short*>(lds + off_pt + ps * 2);
        } else {
            uint2 v = make_uint2(0, 0);
            if (sp && level - cap < sp_levels) { v = sp[spslot(g, level)]; wait_vmem(); }
            w = v.x; t = (int)v.y;
        }
    }
    // POOL == false: every level the scan keeps is private (the host picked cap from the deepest stack of
    // an earlier build of the same data); a level beyond only raises the overflow flag.
    __device__ __forceinline__ void load(int g, int wb, int tb, unsigned tab, int level, unsigned& w, int& t) const {
        if (!POOL || level < cap) { w = W(wb, level); t = T(tb, level); }
        else load_high(g, tab, level, w, t);
    }
    __device__ void store_high(int g, unsigned& tab, int level, unsigned w, int t) const {
        const int L = level - cap, ci = L >> 3;
        if (ci < ENV_MAXCH && ((tab >> (8 * ci)) & 0xffu) == 0xffu) {
            const unsigned id = atomicAdd(counter(), 1u);
            if (id < (unsigned)nch) tab = (tab & ~(0xffu << (8 * ci))) | (id << (8 * ci));
        }
        const int ps = pool_slot(tab, level);
        if (ps >= 0) {
            *reinterpret_cast<unsigned*>(lds + off_pw + ps * 4) = w;
            *reinterpret_cast<unsigned short*>(lds + off_pt + ps * 2) = (unsigned short)t;
        } else if (sp && level - cap < sp_levels) sp[spslot(g, level)] = make_uint2(w, (unsigned)t);   // idle lanes too: same slot, same entry as the row they shadow
        else *ovf = 1;
    }
    __device__ __forceinline__ void store(int g, int wb, int tb, unsigned& tab, int level, unsigned w, int t) const {
        if (level < cap) { W(wb, level) = w; T(tb, level) = (unsigned short)t; }
        else if (POOL) store_high(g, tab, level, w, t);
        else *ovf = 1;
    }
};

// Scan state of one stack: the top entry unpacked, the entry below it packed.
struct EnvTop {
    int q;                  // index of the top entry, -1 = empty
    unsigned tw; int tt2;   // top, packed, and twice its start
    unsigned bw; int bt2;   // entry below the top (undefined when q < 1)
    int wb, tb;             // byte offsets of level 0 of this stack in the workgroup's LDS block
    unsigned tab;           // pool chunk ids of this stack
    int deepest;            // largest q so far
    unsigned aw, at;        // LDS addresses of level -1 (words, starts), for the hand-written pop loop
};

// 32-bit LDS address of a pointer into the workgroup's LDS block
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

template <int G, bool POOL>
__device__ __forceinline__ void env_init(const Env<G, POOL>& E, int g, EnvTop& e) {
    // A sentinel entry sits below level 0 and is what an empty stack holds as its top: site 0 with the largest value
    // there is, from position 0.
    e = EnvTop{-1, ENV_VMASK, 0, 0u, 0, E.wbase(g), E.tbase(g), 0xffffffffu, -1, 0u, 0u};
    e.aw = lds_address(E.lds + e.wb - 256); e.at = lds_address(E.lds + e.tb - 128);
    E.W(e.wb, -1) = ENV_VMASK; E.T(e.tb, -1) = 0;
}

// Adds site u (ascending) with value v to the stack of group g: pop while the new parabola is lower at
// the start t of the top entry, then enter it at the first position where it is lower than the top, if
// that position is inside the row (src/testLDT.cpp:136-150). With a = s^2 + v the parabola of a site is
// f(p) = p^2 - 2 p s + a, so "new lower than top at t" is 2 t (u - s) + (a_top - a_u) > 0 and the
// crossing is (a_u - a_top) / (2 (u - s)); entries carry a instead of v. Reads below the bottom of a
// stack return whatever precedes it in LDS and are never used.
// SAFE (hand-written variant only): the stack cannot outgrow its levels (the group has no more sites than levels),
// so the level needs no clamp and the deepest level is not tracked.
template <int G, bool POOL, bool SAFE = false>
__device__ __forceinline__ void env_add(const Env<G, POOL>& E, int g, EnvTop& e, int u, int v, int div) {
    const int au = (u & 0x3ff) * (u & 0x3ff) + v;
    int d, dd, w;
    // With the pool the hand-written step serves as long as every stack of the wave stays below its last private level
    // (it reads and writes private levels only); a wave with a deeper stack takes the general step for that site.
    if (!POOL || __all(e.q < E.cap - 1)) {
        // The pop loop and the crossing, by hand. The loop runs while any lane of the wave has to pop and is
        // predicated instead of divergent (lanes that are done re-read the entry they already hold): 18
        // instructions per round where the compiled divergent loop needs about 30, most of them mask
        // bookkeeping, and a wave issues one instruction every eight cycles or so whatever its kind. The
        // push follows as straight-line selects. The compiler does not see the LDS traffic in here: the wait for
        // the loop's last read-back sits behind the crossing's arithmetic, which covers most of its latency,
        // and the writes need none (LDS operations of a wave complete in order).
        int x, y;
        asm volatile(
            "Lenv_pop_%=:\n"
            "v_lshrrev_b32 %[d], 22, %[tw]\n"
            "v_and_b32 %[dd], 0x3fffff, %[tw]\n"
            "v_sub_u32 %[d], %[u], %[d]\n"
            "v_sub_u32 %[dd], %[dd], %[au]\n"
            "v_mad_i32_i24 %[x], %[tt2], %[d], %[dd]\n"
            // pop while x > 0 and q >= 0, i.e. min(x, q + 1) > 0: the second condition stops at the sentinel,
            // whatever the input
            "v_add_u32 %[y], 1, %[q]\n"
            "v_min_i32 %[x], %[x], %[y]\n"
            "v_cmp_lt_i32 vcc, 0, %[x]\n"
            "s_cbranch_vccz Lenv_done_%=\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_cndmask_b32 %[tw], %[tw], %[bw], vcc\n"
            "v_cndmask_b32 %[tt2], %[tt2], %[bt2], vcc\n"
            "v_subbrev_co_u32 %[q], vcc, 0, %[q], vcc\n"
            "v_lshl_add_u32 %[x], %[q], 8, %[aw]\n"
            "v_lshl_add_u32 %[y], %[q], 7, %[at]\n"
            "ds_read_b32 %[bw], %[x]\n"
            "ds_read_u16 %[bt2], %[y]\n"
            "s_branch Lenv_pop_%=\n"
            "Lenv_done_%=:\n"
            // crossing: w = 1 + floor((2 N + 1) / (2 D)) with N = -dd, D = 2 d (floordiv_pos)
            "v_mad_i32_i24 %[x], %[dd], -2, 1\n"
            "v_lshlrev_b32 %[y], 2, %[d]\n"
            "v_cvt_f32_i32 %[x], %[x]\n"
            "v_cvt_f32_i32 %[y], %[y]\n"
            "v_rcp_f32 %[y], %[y]\n"
            "s_nop 0\n"
            "v_mul_f32 %[x], %[x], %[y]\n"
            "v_cvt_i32_f32 %[x], %[x]\n"
            "v_add_u32 %[w], 1, %[x]\n"
            // Against the sentinel the quotient is below -500 for every site there can be (the sentinel's value exceeds
            // any site's by more than a million, the divisor is at most 4 * 1022; u = 0 gives -inf, converted to the
            // most negative integer): an empty stack takes the site from position 0 by this clamp alone. Real
            // entries cross at positive positions. The site enters if its first position is inside the row.
            "v_max_i32 %[w], 0, %[w]\n"
            "v_lshl_or_b32 %[x], %[u], 22, %[au]\n"
            "v_lshlrev_b32 %[y], 1, %[w]\n"
            "v_cmp_gt_i32 vcc, %[div], %[w]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_cndmask_b32 %[bw], %[bw], %[tw], vcc\n"
            "v_cndmask_b32 %[bt2], %[bt2], %[tt2], vcc\n"
            "v_cndmask_b32 %[tw], %[tw], %[x], vcc\n"
            "v_cndmask_b32 %[tt2], %[tt2], %[y], vcc\n"
            "v_addc_co_u32 %[q], vcc, 0, %[q], vcc\n"
            // every lane writes its top entry (lanes that did not push rewrite what is there); unless the stack is
            // known to fit, the level is clamped to the last private one and a stack that got there is reported by
            // env_finish
            ".if %c[safe] == 0\n"
            "v_max_i32 %[deep], %[deep], %[q]\n"
            "v_min_i32 %[d], %[capm1], %[q]\n"
            "v_lshl_add_u32 %[dd], %[d], 8, %[aw]\n"
            "v_lshl_add_u32 %[d], %[d], 7, %[at]\n"
            ".else\n"
            "v_lshl_add_u32 %[dd], %[q], 8, %[aw]\n"
            "v_lshl_add_u32 %[d], %[q], 7, %[at]\n"
            ".endif\n"
            "ds_write_b32 %[dd], %[tw] offset:256\n"
            "ds_write_b16 %[d], %[tt2] offset:128\n"
            : [tw] "+v"(e.tw), [tt2] "+v"(e.tt2), [bw] "+v"(e.bw), [bt2] "+v"(e.bt2), [q] "+v"(e.q), [deep] "+v"(e.deepest),
              [d] "=&v"(d), [dd] "=&v"(dd), [x] "=&v"(x), [y] "=&v"(y), [w] "=&v"(w)
            : [u] "s"(u), [au] "v"(au), [aw] "v"(e.aw), [at] "v"(e.at), [div] "s"(div), [capm1] "s"(E.cap - 1), [safe] "n"(SAFE ? 1 : 0)
            : "vcc", "memory");
        return;
    }
    if constexpr (POOL) {
        for (;;) {
            d = u - (int)(e.tw >> ENV_VB); dd = (int)(e.tw & ENV_VMASK) - au;
            if (!(e.q >= 0 && __mul24(e.tt2, d) + dd > 0)) break;
            e.tw = e.bw; e.tt2 = e.bt2;
            e.q--;
            E.load(g, e.wb, e.tb, e.tab, e.q - 1, e.bw, e.bt2);
        }
        w = 1 + floordiv_pos(-dd, 2 * d);
    }
    if (e.q < 0) w = 0;
    if (w < div) {
        e.bw = e.tw; e.bt2 = e.tt2;
        e.tw = ((unsigned)u << ENV_VB) | (unsigned)au; e.tt2 = 2 * w;
        e.q++;
        e.deepest = max(e.deepest, e.q);
        E.store(g, e.wb, e.tb, e.tab, e.q, e.tw, e.tt2);
    }
}

// Ends the scan of group g: publishes depth and chunk table (a group without sites gets one entry that
// loses against every real one) and, when max_depth is not NULL, reports the deepest level count any stack
// of the wave reached, which the host uses to size the private levels of later builds.
template <int G, bool POOL>
__device__ __forceinline__ void env_finish(const Env<G, POOL>& E, int g, EnvTop& e, int deepest, u64* max_depth) {
    for (int o = 32; o; o >>= 1) deepest = max(deepest, __shfl_xor(deepest, o));
    if (max_depth && E.lane == 0 && (u64)deepest > *(volatile u64*)max_depth) atomicMax(max_depth, (u64)deepest);
    if (!POOL && e.deepest >= E.cap) *E.ovf = 1;   // the hand-written scan clamps instead of checking every push
    if (e.q < 0) { e.q = 0; E.store(g, e.wb, e.tb, e.tab, 0, ENV_VMASK, 0); }
    E.depth(g) = (unsigned short)e.q;
    if (POOL) E.table(g) = e.tab;
}

// A wave's 64 lanes read or write one 256-byte piece of a grid row: base (the same in all lanes) + a scalar byte offset
// (which row) + 4 * lane_off, lane_off < 64. As a buffer access the base and the row offset stay in scalar registers (one
// scalar addition per step); as a flat pointer the compiler keeps a 64-bit address per lane and spends vector
// instructions on it at every step, and a base rebuilt for every row costs half a dozen scalar instructions, which in
// the read-back loop (seven vector instructions per position) is what the SIMD then waits for. The byte offset of a
// row is below 2^32 for every grid the library accepts (1022 * 1023^2 * 4). No range check is wanted (all lanes are in
// range by construction): the descriptor covers the whole address range above its base.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t grid_rows(const int* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ void row_store(__amdgpu_buffer_rsrc_t rows, unsigned row_bytes, unsigned lane_off, int v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, rows, (int)(lane_off * 4u), (int)row_bytes, 0);
}
__device__ __forceinline__ int row_load(__amdgpu_buffer_rsrc_t rows, unsigned row_bytes, unsigned lane_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(rows, (int)(lane_off * 4u), (int)row_bytes, 0);
}

// Evaluation cursor on one stack, walking towards position 0.
struct EnvCur {
    int lvl;                // level of the current entry
    unsigned cw; int ct2;   // current entry (start doubled, as stored)
    unsigned bw; int bt2;   // entry one level below (whatever precedes the stack in LDS when lvl == 0: never used)
    int wb, tb;             // LDS byte addresses of level 0 of the stack
    unsigned tab;
};

// Positions the cursor on the entry of group j that owns position p (largest level whose start is <= p).
template <int G, bool POOL>
__device__ __forceinline__ void env_seek(const Env<G, POOL>& E, int j, EnvCur& c, int p) {
    c.wb = E.wbase(j); c.tb = E.tbase(j); c.tab = POOL ? E.table(j) : 0xffffffffu;
    int lo = 0, hi = E.depth(j);
    while (lo < hi) {
        const int m = (lo + hi + 1) >> 1;
        unsigned w; int t2;
        E.load(j, c.wb, c.tb, c.tab, m, w, t2);
        if (t2 <= 2 * p) lo = m; else hi = m - 1;
    }
    c.lvl = lo;
    E.load(j, c.wb, c.tb, c.tab, lo, c.cw, c.ct2);
    E.load(j, c.wb, c.tb, c.tab, lo - 1, c.bw, c.bt2);
}

// f(p) - p^2 of the entry under the cursor: a 24-bit multiply-add
__device__ __forceinline__ int env_value(const EnvCur& c, int p) {
    // cw = s * 2^22 + a, so a - 2 p s = cw - s (2^22 + 2 p) in 32-bit arithmetic: no mask needed (0 <= p <= 1022, the
    // multiplier stays inside 24 bits)
    return (int)((unsigned)__mul24((int)(c.cw >> ENV_VB), -(2 * p + (1 << ENV_VB))) + c.cw);
}

// One level down if p was the first position of the current entry. The entry below is already in
// registers; the one below that is read back every time (branch-free) and is not needed before the next
// step of this cursor, so the LDS latency stays off the dependent chain.
// HIGH: some lane's cursor may still be above the private levels.
template <int G, bool POOL, bool HIGH>
__device__ __forceinline__ void env_step(const Env<G, POOL>& E, int j, EnvCur& c, int p) {
    const bool down = 2 * p == c.ct2;
    c.cw = down ? c.bw : c.cw;
    c.ct2 = down ? c.bt2 : c.ct2;
    c.lvl -= down ? 1 : 0;
    if (HIGH) E.load(j, c.wb, c.tb, c.tab, c.lvl - 1, c.bw, c.bt2);
    else { c.bw = E.W(c.wb - 256, c.lvl); c.bt2 = E.T(c.tb - 128, c.lvl); }
}

// The step for cursors inside the private levels: the current entry is read again at every position instead of being
// selected from a prefetched entry below. Two selects fewer per stack and position; the LDS read is on the cursor's
// dependent chain, which the other waves of the SIMD cover (the read-back is bound by instruction issue: pass x
// 44.8 -> 43.2 us, pass zy 25.3 -> 23.7 us against the prefetching step).
template <int G, bool POOL>
__device__ __forceinline__ void env_step_reload(const Env<G, POOL>& E, EnvCur& c, int p) {
    c.lvl -= (2 * p == c.ct2) ? 1 : 0;
    c.cw = E.W(c.wb, c.lvl); c.ct2 = E.T(c.tb, c.lvl);
}

// Positions [plo, p] of this lane's row from stacks LO .. LO + N - 1 only.
template <int G, bool POOL, int LO, int N>
__device__ __forceinline__ void env_span(const Env<G, POOL>& E, EnvCur (&cur)[G], int* __restrict__ out, size_t stride, unsigned lane_off,
                                         int plo, int p) {
    const __amdgpu_buffer_rsrc_t rows = grid_rows(out);
    const unsigned stride4 = (unsigned)stride * 4u;
    for (; POOL;) {   // general steps while any cursor of the wave is above the private levels
        bool high = false;
#pragma unroll
        for (int j = LO; j < LO + N; j++) high |= cur[j].lvl >= E.cap;
        if (p < plo || !__any(high)) break;
        int v = 0x7fffffff;
#pragma unroll
        for (int j = LO; j < LO + N; j++) v = min(v, env_value(cur[j], p));
        row_store(rows, (unsigned)p * stride4, lane_off, v + p * p);   // idle lanes shadow the last row: same value, same address
#pragma unroll
        for (int j = LO; j < LO + N; j++) {   // only the stacks that are still above pay for the general read-back
            if (__any(cur[j].lvl >= E.cap)) env_step<G, POOL, true>(E, j, cur[j], p);
            else env_step<G, POOL, false>(E, j, cur[j], p);
        }
        p--;
    }
#pragma unroll 2
    for (; p >= plo; p--) {
        int v = 0x7fffffff;
#pragma unroll
        for (int j = LO; j < LO + N; j++) v = min(v, env_value(cur[j], p));
        row_store(rows, (unsigned)p * stride4, lane_off, v + p * p);   // idle lanes shadow the last row: same value, same address
#pragma unroll
        for (int j = LO; j < LO + N; j++) env_step_reload<G, POOL>(E, cur[j], p);
    }
}

template <int G, bool POOL, int LO, int N>
__device__ __forceinline__ void env_dispatch(int jlo, int n, const Env<G, POOL>& E, EnvCur (&cur)[G], int* __restrict__ out, size_t stride,
                                             unsigned lane_off, int plo, int p) {
    if constexpr (LO + N <= G) {
        if (jlo == LO && n == N) { env_span<G, POOL, LO, N>(E, cur, out, stride, lane_off, plo, p); return; }
        if constexpr (LO + N < G) env_dispatch<G, POOL, LO, N + 1>(jlo, n, E, cur, out, stride, lane_off, plo, p);
        else if constexpr (LO + 1 < G) env_dispatch<G, POOL, LO + 1, 1>(jlo, n, E, cur, out, stride, lane_off, plo, p);
    }
}

// Writes positions [plo, phi) of this lane's row: out[p * stride + lane_off] = min over the G stacks;
// wave w of the workgroup handles part w of `parts` parts of the positions (parts = number of waves: the G
// that scanned, plus, in pass zy, as many again that only help here). Called by all waves (contains a barrier).
// Stack i holds sites left of those of stack j > i, so envelope_i(p) - envelope_j(p) grows with p and the
// index of the stack that attains the minimum never decreases with p: the stacks that can matter inside
// a part lie between a winner at the last position of the previous part (found by the previous wave) and
// a winner at the part's own last position. The wave evaluates only the range of stacks that covers this
// interval for all of its lanes (usually one or two of the G).
template <int G, bool POOL>
__device__ __forceinline__ void env_write(const Env<G, POOL>& E, int g, int parts, int* __restrict__ out, size_t stride, unsigned lane_off, int div) {
    const int plo = div * g / parts, phi = div * (g + 1) / parts;
    EnvCur cur[G];
    int jlo = 0, jhi = G - 1;
    if (G > 1) {
        int vtop[G];
#pragma unroll
        for (int j = 0; j < G; j++) {
            env_seek(E, j, cur[j], max(phi - 1, 0));
            vtop[j] = env_value(cur[j], max(phi - 1, 0));
        }
        int mt = vtop[0];
#pragma unroll
        for (int j = 1; j < G; j++) mt = min(mt, vtop[j]);
        int a = 0;                            // last winner at the top position of this part
#pragma unroll
        for (int j = 0; j < G; j++) { if (vtop[j] == mt) a = j; }
        E.winner(g) = (unsigned char)a;
        __syncthreads();
        const int b = g > 0 ? min((int)E.winner(g - 1), a) : 0;
        jhi = 0; jlo = G - 1;
#pragma unroll
        for (int j = 0; j < G; j++) {
            if (__any(a == j)) jhi = j;
            if (__any(b == G - 1 - j)) jlo = G - 1 - j;
        }
    } else {
        env_seek(E, 0, cur[0], max(phi - 1, 0));
    }
    if (phi <= plo) return;
    env_dispatch<G, POOL, 0, 1>(jlo, jhi - jlo + 1, E, cur, out, stride, lane_off, plo, phi - 1);
}

// LDS layout: header (depths, chunk tables, pool counter, winners of up to 2 G parts) padded so that reads two
// levels below level 0 of group 0 stay inside the allocation, then the private levels, then the pool.
// parts: waves of the workgroup (one winner byte per lane each, behind the 64-byte counter slot)
// Without the pool the header is just depths and winners (no chunk tables, no counter).
__host__ __device__ inline int env_off_w(int G, int parts, bool pool) {
    if (!pool) { const int h = G * 128 + parts * 64; return h < 512 ? 512 : h; }
    const int pad = (64 + parts * 64 + 255) & ~255;
    return G * 384 + (pad < 512 ? 512 : pad);
}
static size_t env_lds_bytes(int G, int parts, int cap, int nch, bool pool) { return (size_t)env_off_w(G, parts, pool) + (size_t)G * (cap + 1) * 384 + (size_t)nch * ENV_CHUNK * 6; }

template <int G, bool POOL>
__device__ __forceinline__ Env<G, POOL> env_setup(unsigned* lds, int parts, int cap, int nch, int lane, uint2* sp, int div, size_t rows, size_t row,
                                            u64* ovf) {
    Env<G, POOL> E;
    E.lds = reinterpret_cast<char*>(lds);
    E.cap = cap; E.nch = nch; E.lane = lane;
    E.off_w = env_off_w(G, parts, POOL);
    E.off_t = E.off_w + G * E.rows() * 256;
    E.off_pw = E.off_t + G * E.rows() * 128;
    E.off_pt = E.off_pw + nch * ENV_CHUNK * 4;
    E.sp = sp ? sp + row : nullptr;
    E.sp_levels = max((div + G - 1) / G + 1 - cap, 1);   // the host sizes the buffer with the same rule
    E.sp_gstride = (unsigned)rows; E.sp_stride = (unsigned)rows * G; E.ovf = ovf;
    if (POOL) {
        if (threadIdx.x == 0) *E.counter() = 0;
        __syncthreads();
    }
    return E;
}

// ---- pass zy ------------------------------------------------------------------------------------------
// Workgroup = plane x, 64 consecutive z; lane = z. Sites are the plane's non-empty rows y; their records
// are fetched 64 at a time (lane j holds record j of the wave's site range, one coalesced load, the next
// block prefetched) and broadcast with v_readlane as the scan reaches them. Value = (distance along z)^2.
// RB = 2: as many waves again join after the scan and take half of the read-back positions each; they wait
// at the barrier meanwhile. A workgroup lasts as long as one wave's scan plus read-back, and there are too
// few workgroups (here: non-empty planes x chunks) for other waves to fill the SIMDs in the meantime, so a
// shorter chain is a shorter pass.
template <int G, bool POOL, int RB>
__global__ __launch_bounds__(64 * G * RB) void k_edt_pass_zy(const RowRec* __restrict__ recs, const int* __restrict__ rowcnt,
                                                        int* __restrict__ plane, uint2* __restrict__ spill_buf, u64* __restrict__ counters,
                                                        int div, int zs, int W, int cap, int nch, u64* __restrict__ next_counters, int padded,
                                                        int* __restrict__ planelist, int report) {
    extern __shared__ unsigned lds[];
    if (blockIdx.x == (unsigned)(div * W)) {   // the extra workgroup: the plane list for pass x
        if (threadIdx.x < 64) edt_list_planes(rowcnt, div, padded, planelist, counters, next_counters, (int)threadIdx.x);
        return;
    }
    const int x = blockIdx.x / W, c = blockIdx.x % W;
    const int cnt = rowcnt[x];
    if (cnt == 0) return;   // plane without voxels: never read by pass x
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z = min(c * 64 + lane, div - 1);   // lanes beyond the row's end shadow its last voxel: same work, same stores
    const int zl = z & 63;   // the lane's offset inside the chunk: addresses are (uniform base) + 4 zl, which lets the
                             // loads and stores take their base from scalar registers
    const Env<G, POOL> E = env_setup<G, POOL>(lds, G * RB, cap, nch, lane, spill_buf, div, (size_t)div * div, (size_t)x * div + z, &counters[4]);

    if (g < G) {
        const int klo = cnt * g / G, khi = cnt * (g + 1) / G;
        const int4* rr4 = reinterpret_cast<const int4*>(recs + ((size_t)x * W + c) * div);
        EnvTop e;
        env_init(E, g, e);
        int4 nxt = klo + lane < khi ? rr4[klo + lane] : make_int4(0, 0, 0, 0);
        auto scan = [&](auto safe) {
        for (int k0 = klo; k0 < khi; k0 += 64) {
            const int4 cur = nxt;
            if (k0 + 64 + lane < khi) nxt = rr4[k0 + 64 + lane];
            const int m = min(64, khi - k0);
            for (int j = 0; j < m; j++) {
                const unsigned lo = (unsigned)bcast(cur.x, j), hi = (unsigned)bcast(cur.y, j);
                const unsigned ba = (unsigned)bcast(cur.z, j);
                const int u = bcast(cur.w, j);
                const u64 own = ((u64)hi << 32) | lo;
                const int below = (int)(ba & 0xffff) - 1, above = (int)(ba >> 16);
                // nearest voxel of the row outside this chunk (the record is the same for all lanes: scalar work)
                const int far_below = below >= 0 ? below : -BIGD, far_above = above != 0xffff ? above : BIGD;
                int gd = min(z - far_below, far_above - z);
                if (own != 0) {   // wave-uniform: most rows of a plane have no voxel inside a given chunk
                    const u64 md = own & (~0ull >> (63 - zl));
                    const u64 mu = own >> zl;
                    if (md) gd = min(gd, zl - (63 - __clzll(md)));   // a voxel inside the chunk is nearer than any beyond it on its side
                    if (mu) gd = min(gd, __ffsll((long long)mu) - 1);
                }
                env_add<G, POOL, decltype(safe)::value>(E, g, e, u, __mul24(gd, gd), div);
            }
        }
        };
        // a group with no more sites than levels cannot overflow its stack: the cheaper site step
        if (!POOL && !report && khi - klo <= cap) scan(std::true_type{}); else scan(std::false_type{});
        env_finish(E, g, e, e.deepest + 1, report ? &counters[3] : nullptr);
    }
    __syncthreads();
    env_write(E, g, G * RB, plane + (size_t)x * div * zs + c * 64, (size_t)zs, (unsigned)zl & 63u, div);
}

// ---- pass x ---------------------------------------------------------------------------------------------
// Workgroup = row y, 64 consecutive z; lane = z. Sites are the non-empty planes; wave g scans the g-th
// part of the plane list, fetching the in-plane distances EDT_PF planes ahead (the list is padded, so
// the group loads need no bounds checks; indices are wave-uniform scalar loads, values one coalesced load
// per plane). Measured on the 300^3 bunny grid: 50.2 / 47.2 / 47.5 / 47.7 / 48.6 / 52.7 us for 1 / 2 / 3 / 4 / 8 / 16
// planes ahead without the pool (three workgroups a CU cover the latency between them; deeper prefetch only costs
// registers); the pool variant, which the larger grids run, does better with eight (768^3: 1.08 against 1.18 ms a build).
template <bool POOL> constexpr int EDT_PF = POOL ? 8 : 2;
// Read-back helpers of the default shape (4 site groups): 2 = as many waves again. Measured on the 300^3
// bunny grid: pass zy 34.8 / 30.3 / 31.5 / 32.1 us and pass x 65.4 / 58.9 / 70.7 / 73.0 us for 1 / 2 / 3 / 4
// (from 3 on a CU no longer holds three workgroups of pass x).
constexpr int EDT_ZY_RB = 2, EDT_X_RB = 2;

template <int G, bool POOL, int RB>
__global__ __launch_bounds__(64 * G * RB) void k_edt_pass_x(const int* __restrict__ plane, const int* __restrict__ planelist,
                                                       u64* __restrict__ counters, int* __restrict__ sq, uint2* __restrict__ spill_buf,
                                                       int div, int zs, int nchunk, int cap, int nch, int report) {
    constexpr int PF = EDT_PF<POOL>;
    extern __shared__ unsigned lds[];
    // Workgroup ids go round the 8 XCDs: with a chunk count that is a multiple of 8 a plain numbering gives every
    // XCD the same one or two z chunks of all rows, and the chunks differ a lot in work (bunny, 1008^3: 2.63 ms
    // against 1.87 ms with the chunks rotated by row). Other counts rotate by themselves; at 300^3 (5 chunks) the
    // rotation costs 2 us, so it is applied only where it is needed.
    // pass zy overflowed its stacks: the plane values are not all written, the build is repeated with other shapes
    // anyway, and nothing this pass could compute from them would be used
    if (counters[4] != 0) return;
    const int y = blockIdx.x / nchunk;
    int c = blockIdx.x % nchunk;
    if ((nchunk & 7) == 0) c = (c + y) % nchunk;
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z = min(c * 64 + lane, div - 1);   // lanes beyond the row's end shadow its last voxel: same work, same stores
    const int np = (int)counters[2];
    const size_t xs = (size_t)div * zs;                   // elements between consecutive x (rows are zs values apart)
    const int rowoff = y * div + z;                       // the row's number (spill buffer)
    const size_t rowbase = (size_t)y * zs + c * 64;       // uniform part of the row's place in a plane
    const unsigned zl = (unsigned)(z - c * 64) & 63u;   // the lane's part
    const Env<G, POOL> E = env_setup<G, POOL>(lds, G * RB, cap, nch, lane, spill_buf, div, (size_t)div * div, (size_t)rowoff, &counters[5]);

    if (g < G) {
        const int klo = np * g / G, khi = np * (g + 1) / G;
        EnvTop e;
        env_init(E, g, e);
        int Gn[PF], Un[PF];
        const __amdgpu_buffer_rsrc_t planes = grid_rows(plane + rowbase);
        const unsigned xs4 = (unsigned)xs * 4u;
    #pragma unroll
        for (int j = 0; j < PF; j++) Un[j] = planelist[klo + j];
    #pragma unroll
        for (int j = 0; j < PF; j++) Gn[j] = row_load(planes, (unsigned)Un[j] * xs4, zl);
        auto scan = [&](auto safe) {
        for (int k0 = klo; k0 < khi; k0 += PF) {
            int Gc[PF], Uc[PF];
    #pragma unroll
            for (int j = 0; j < PF; j++) { Gc[j] = Gn[j]; Uc[j] = Un[j]; }
    #pragma unroll
            for (int j = 0; j < PF; j++) Un[j] = planelist[k0 + PF + j];
    #pragma unroll
            for (int j = 0; j < PF; j++) Gn[j] = row_load(planes, (unsigned)Un[j] * xs4, zl);
    #pragma unroll
            for (int j = 0; j < PF; j++)
                if (k0 + j < khi) env_add<G, POOL, decltype(safe)::value>(E, g, e, Uc[j], Gc[j], div);
        }
        };
        if (!POOL && !report && khi - klo <= cap) scan(std::true_type{}); else scan(std::false_type{});
        env_finish(E, g, e, e.deepest + 1, report ? &counters[7] : nullptr);
    }
    __syncthreads();
    env_write(E, g, G * RB, sq + rowbase, xs, zl, div);
}

// ---- lookup (device function in icpx_edt_dev.h) ----
__global__ void k_edt_lookup(const int* __restrict__ sq, int div, int zs, float mnx, float mny, float mnz, float cell,
                             const float* __restrict__ q, size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = edt_distance(sq, div, zs, mnx, mny, mnz, cell, q[3 * i], q[3 * i + 1], q[3 * i + 2]);
}

// DT3D::Distance (src/jly_3ddt.cpp:821-856) on the exact grid: rounded voxel index; inside the grid the
// stored distance sqrt(sq) / scale as float; outside, the clamped voxel's distance plus the overshoot in
// cells (float a, b, c; their squares summed in float; sqrt and division in double) over scale.
__global__ void k_edt_lookup_round(const int* __restrict__ sq, int div, int zs, double mnx, double mny, double mnz, double scale,
                                   const double* __restrict__ q, size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x = (int)((q[3 * i] - mnx) * scale + 0.5);
    int y = (int)((q[3 * i + 1] - mny) * scale + 0.5);
    int z = (int)((q[3 * i + 2] - mnz) * scale + 0.5);
    float a = 0.f, b = 0.f, c = 0.f;
    if (x < 0) { a = (float)x; x = 0; } else if (x >= div) { a = (float)(x - div + 1); x = div - 1; }
    if (y < 0) { b = (float)y; y = 0; } else if (y >= div) { b = (float)(y - div + 1); y = div - 1; }
    if (z < 0) { c = (float)z; z = 0; } else if (z >= div) { c = (float)(z - div + 1); z = div - 1; }
    const float g = (float)(__dsqrt_rn((double)sq[((size_t)x * div + y) * zs + z]) / scale);
    if (a == 0.f && b == 0.f && c == 0.f) { out[i] = g; return; }
    const float s2 = __fadd_rn(__fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b)), __fmul_rn(c, c));
    out[i] = (float)(__dsqrt_rn((double)s2) / scale + (double)g);
}

}  // namespace

// Two sets of uint64 x 8 behind the occupancy bits, used by alternate builds: {occupied voxels, dropped
// points, non-empty planes, deepest stack of pass zy, overflow pass zy, overflow pass x, unused,
// deepest stack of pass x}
static u64* edt_counter_set(const icpx_ctx* ctx, int set) {
    return ctx->edt_bits.as<u64>() + (size_t)ctx->div * ctx->div * ctx->edt_words + 8 * (set & 1);
}
static u64* edt_counters(const icpx_ctx* ctx) { return edt_counter_set(ctx, ctx->edt_epoch); }

// Size of the spill buffer of the envelope passes at the current resolution: one 8-byte entry per row, site
// group and level beyond the private ones (a group holds at most ceil(div / groups) sites).
static size_t edt_spill_bytes(const icpx_ctx* ctx) {
    size_t need = 0;
    for (const EnvTune* t : {&ctx->edt_zy, &ctx->edt_x}) {
        const int levels = std::max(((int)ctx->div + t->groups - 1) / t->groups + 1 - t->cap, 1);
        need = std::max(need, (size_t)levels * t->groups * ctx->div * ctx->div * sizeof(uint2));
    }
    return need;
}

// Shapes the LDS of the envelope passes for later builds from the deepest stacks the last build saw: if
// they fit a workgroup's private levels at three workgroups per CU (pass x: two), every level is private and
// the pool is dropped (reads and writes above the private levels cost several times more); otherwise the default
// private levels plus pool stay. Shapes forced through the environment are left alone.
static void edt_adapt(icpx_ctx* ctx, u64 deepest_zy, u64 deepest_x) {
    if (ctx->edt_tune_fixed) return;
    const EnvTune def{4, 28, 130};
    // 4 groups x (33 + 1 sentinel) rows x 384 B + 1 KiB header = 52 KiB: three workgroups per CU. Pass x still gains
    // from private levels alone at two workgroups per CU, up to 40 levels (64 000 B; bunny at 320^3 to 360^3,
    // stacks of 35 to 38: 0.070 / 0.087 / 0.098 ms against 0.080 / 0.098 / 0.109 with the pool); pass zy does not.
    const int cap_max[2] = {33, 40};
    const u64 d[2] = {deepest_zy, deepest_x};
    EnvTune* t[2] = {&ctx->edt_zy, &ctx->edt_x};
    bool fits = true;
    for (int k = 0; k < 2; k++) {
        const bool f = d[k] >= 1 && d[k] <= (u64)cap_max[k];
        *t[k] = f ? EnvTune{4, std::max((int)d[k], 8), 0} : def;
        fits = fits && f;
    }
    if (fits) ctx->edt_overflowed = false;
    if (getenv("ICPX_EDT_VERBOSE"))
        fprintf(stderr, "icpx edt: div %u deepest stack zy %llu x %llu -> levels,chunks zy %d,%d x %d,%d\n", ctx->div, (unsigned long long)deepest_zy,
                (unsigned long long)deepest_x, ctx->edt_zy.cap, ctx->edt_zy.chunks, ctx->edt_x.cap, ctx->edt_x.chunks);
}

// Launch on the stream; with ev != NULL the dispatch carries the start / stop events of its stage, which
// time the kernel itself (no extra packets on the stream).
template <class K, class... A>
static void edt_launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipEvent_t* ev, int stage, A... args) {
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, ev[2 * stage], ev[2 * stage + 1], 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, args...);
}

// ev: NULL, or 2 * ICPX_EDT_NSTAGES events, a (start, stop) pair per stage: splat, census, pass zy, pass x.
// report: the passes also record the deepest stack they saw (read back by the checked builds, which shape
// the LDS of later ones with it; 6000 waves polling one counter cost 12 us per build, so rebuilds of the
// same data skip it).
static int edt_run_impl(icpx_ctx* ctx, hipEvent_t* ev, int report) {
    const int div = (int)ctx->div, W = ctx->edt_words;
    const int nchunk = (div + 63) / 64;
    hipStream_t s = ctx->stream;
    // A spill buffer left from a build at a lower resolution is too small for this one: run without it
    // (the build then reports the overflow, if any, and the buffer is grown).
    uint2* spill = ctx->edt_spill.cap >= edt_spill_bytes(ctx) ? ctx->edt_spill.as<uint2>() : nullptr;
    // No clearing here: the census of the previous build left the occupancy bits zero and zeroed this
    // build's counter set (edt_prepare clears everything when the buffers are (re)allocated or resized).
    ctx->edt_epoch ^= 1;
    ctx->edt_bricks_ready = false;   // the bound kernel's brick copy belongs to the previous grid
    u64* counters = edt_counters(ctx);
    if (ctx->edt_round)
        edt_launch(k_edt_splat_round, dim3((unsigned)((ctx->nt + 255) / 256)), dim3(256), 0, s, ev, 0, (const float*)ctx->tgt_xyz.as<float>(),
                   ctx->nt, ctx->edt_dmin[0], ctx->edt_dmin[1], ctx->edt_dmin[2], ctx->edt_scale, div, W, ctx->edt_bits.as<u64>(), counters);
    else
        edt_launch(k_edt_splat, dim3((unsigned)((ctx->nt + 255) / 256)), dim3(256), 0, s, ev, 0, (const float*)ctx->tgt_xyz.as<float>(), ctx->nt,
                   ctx->edt_min[0], ctx->edt_min[1], ctx->edt_min[2], ctx->edt_cell, div, W, ctx->edt_bits.as<u64>(), counters);
    edt_launch(k_edt_census, dim3(div), dim3(64 * nchunk), 0, s, ev, 1, ctx->edt_bits.as<u64>(), div, W, ctx->edt_rowlist.as<RowRec>(),
               ctx->edt_rowcnt.as<int>(), counters);
    {
        const int G = ctx->edt_zy.groups, cap = ctx->edt_zy.cap, nch = ctx->edt_zy.chunks;
        const bool pool = nch > 0 || ctx->edt_overflowed;
        const int RB = G == 4 ? EDT_ZY_RB : 1;   // read-back helpers for the default shape only
        auto zy = pool ? (G == 1 ? k_edt_pass_zy<1, true, 1> : G == 2 ? k_edt_pass_zy<2, true, 1> : G == 3 ? k_edt_pass_zy<3, true, 1> : k_edt_pass_zy<4, true, EDT_ZY_RB>)
                       : (G == 1 ? k_edt_pass_zy<1, false, 1> : G == 2 ? k_edt_pass_zy<2, false, 1> : G == 3 ? k_edt_pass_zy<3, false, 1> : k_edt_pass_zy<4, false, EDT_ZY_RB>);
        edt_launch(zy, dim3(div * nchunk + 1), dim3(64 * G * RB), env_lds_bytes(G, G * RB, cap, nch, pool), s, ev, 2, (const RowRec*)ctx->edt_rowlist.as<RowRec>(),
                   (const int*)ctx->edt_rowcnt.as<int>(), ctx->edt_plane.as<int>(), spill, counters, div, (int)ctx->edt_zs, W, cap, nch,
                   edt_counter_set(ctx, ctx->edt_epoch ^ 1), (div + 15) / 16 * 16 + 16, ctx->edt_planelist.as<int>(), report);
    }
    {
        const int G = ctx->edt_x.groups, cap = ctx->edt_x.cap, nch = ctx->edt_x.chunks;
        const bool pool = nch > 0 || ctx->edt_overflowed;
        auto px = pool ? (G == 1 ? k_edt_pass_x<1, true, 1> : G == 2 ? k_edt_pass_x<2, true, 1> : G == 3 ? k_edt_pass_x<3, true, 1> : k_edt_pass_x<4, true, EDT_X_RB>)
                       : (G == 1 ? k_edt_pass_x<1, false, 1> : G == 2 ? k_edt_pass_x<2, false, 1> : G == 3 ? k_edt_pass_x<3, false, 1> : k_edt_pass_x<4, false, EDT_X_RB>);
        edt_launch(px, dim3(div * nchunk), dim3(64 * G * (G == 4 ? EDT_X_RB : 1)), env_lds_bytes(G, G * (G == 4 ? EDT_X_RB : 1), cap, nch, pool), s, ev, 3, (const int*)ctx->edt_plane.as<int>(),
                   (const int*)ctx->edt_planelist.as<int>(), counters, ctx->edt_sq.as<int>(), spill, div, (int)ctx->edt_zs, nchunk, cap, nch, report);
    }
    return icpx_launch_check(ctx, "EDT kernels");
}

int icpx_edt_run(icpx_ctx* ctx, bool report) { return edt_run_impl(ctx, nullptr, report ? 1 : 0); }

extern "C" int icpx_edt_rebuild_timed(icpx_ctx* ctx, int reps, int sample_every, float stage_ms[ICPX_EDT_NSTAGES], int* sampled) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_rebuild_timed: no EDT has been built yet");
    if (reps < 1 || sample_every < 1 || !stage_ms || !sampled) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_rebuild_timed: bad arguments");
    ICPX_TRY(icpx_bind(ctx));
    const int ns = (reps + sample_every - 1) / sample_every;
    std::vector<hipEvent_t> ev((size_t)ns * 2 * ICPX_EDT_NSTAGES);
    for (auto& e : ev) ICPX_CHECK_HIP(ctx, hipEventCreate(&e));
    int rc = ICPX_OK;
    for (int r = 0; r < reps && rc == ICPX_OK; r++)
        rc = edt_run_impl(ctx, r % sample_every == 0 ? ev.data() + (size_t)(r / sample_every) * 2 * ICPX_EDT_NSTAGES : nullptr, 0);
    if (rc == ICPX_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e)); rc = ICPX_EHIP; }
    }
    for (int k = 0; k < ICPX_EDT_NSTAGES; k++) stage_ms[k] = 0.f;
    for (int r = 0; r < ns && rc == ICPX_OK; r++)
        for (int k = 0; k < ICPX_EDT_NSTAGES; k++) {
            float ms = 0.f;
            const hipEvent_t* e2 = &ev[((size_t)r * ICPX_EDT_NSTAGES + k) * 2];
            (void)hipEventElapsedTime(&ms, e2[0], e2[1]);
            stage_ms[k] += ms;
        }
    *sampled = ns;
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
    // Rows of the two grids (z runs fastest) start on a 128-byte line when the side is a multiple of 32. Up to some 400
    // cells a side both grids live in the Infinity Cache and it makes no difference (288 / 300 / 304: pass x 46 / 48 /
    // 49 us); a grid of gigabytes pays for every row that straddles a line in HBM (bunny045 with rows back to back:
    // 1008^3 2.2 ms, 1020^3 3.6 ms, 1023^3 3.7 ms; all three 2.04 ms with rows 1024 values apart), so from 512 on the rows
    // are zs = div rounded up to 32 values apart.
    ctx->edt_zs = div >= 512 ? (div + 31) / 32 * 32 : div;
    const size_t n3 = (size_t)div * div * ctx->edt_zs;
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_bits, ((size_t)div * div * ctx->edt_words + 16) * sizeof(u64)));   // + two counter sets
    ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->edt_bits.p, 0, ((size_t)div * div * ctx->edt_words + 16) * sizeof(u64), ctx->stream));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_rowlist, (size_t)div * div * ctx->edt_words * 16));   // RowRec records
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_rowcnt, (size_t)div * sizeof(int)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_planelist, (size_t)((div + 15) / 16 * 16 + 16) * sizeof(int)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_plane, n3 * sizeof(int)));
    ICPX_TRY(icpx_reserve(ctx, ctx->edt_sq, n3 * sizeof(int)));
    return ICPX_OK;
}

// The first build at a resolution up to this one is tried with private levels only (EDT_FIRST_CAP of them, the most
// that keeps three workgroups per CU) and without the deepest-stack report: a surface scan's stacks at 300^3 are 28 / 33
// levels deep, and such a build then costs what the adapted rebuilds cost (pass x at 300^3: 43 us against 74 us with
// the pool variant and the report, which 6 000 waves pay for by polling one counter). A cloud whose stacks are deeper
// raises the overflow flag (the no-pool site step clamps its level and env_finish reports it) and the build is repeated
// with the pool variant, which is what larger grids start with: their stacks hardly ever fit, and a wasted attempt
// costs a whole build.
constexpr uint32_t EDT_FIRST_DIV = 320;
constexpr int EDT_FIRST_CAP = 33;

// Runs the kernels until a build fits its stack storage (private levels only where that is likely to do, then the
// pool, then pool and spill buffer) and checks that the grid is not empty.
static int edt_build_attempts(icpx_ctx* ctx) {
    const EnvTune def{4, 28, 130};
    if (!ctx->edt_tune_fixed && ctx->edt_shape_div != ctx->div) {
        ctx->edt_zy = ctx->edt_x = ctx->div <= EDT_FIRST_DIV ? EnvTune{4, EDT_FIRST_CAP, 0} : def;
        ctx->edt_overflowed = false;
    }
    ctx->edt_shape_div = ctx->div;
    // Private levels only (the first choice above, or what an earlier build's report led to): no report, a deeper
    // stack shows as an overflow and sends the build to the pool variant
    const bool optimistic = !ctx->edt_tune_fixed && !ctx->edt_overflowed && ctx->edt_zy.chunks == 0 && ctx->edt_x.chunks == 0;
    hipEvent_t ev[2 * ICPX_EDT_NSTAGES];
    if (ctx->edt_time_builds)
        for (auto& e : ev) ICPX_CHECK_HIP(ctx, hipEventCreate(&e));
    int rc = ICPX_OK;
    for (int attempt = 0; attempt < 3; attempt++) {
        const bool first_try = optimistic && attempt == 0;
        // this attempt runs the kernels that use the spill buffer, and has one of the right size
        const bool with_spill = ctx->edt_overflowed && ctx->edt_spill.cap >= edt_spill_bytes(ctx);
        rc = edt_run_impl(ctx, ctx->edt_time_builds ? ev : nullptr, first_try ? 0 : 1);
        if (rc != ICPX_OK) break;
        u64 counters[8];
        hipError_t he = hipMemcpyAsync(counters, edt_counters(ctx), sizeof counters, hipMemcpyDeviceToHost, ctx->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
        if (he != hipSuccess) { ctx->set_error("EDT build: reading the counters failed: %s", hipGetErrorString(he)); rc = ICPX_EHIP; break; }
        ctx->edt_build_attempts_made = attempt + 1;
        if (counters[0] == 0) { ctx->set_error("EDT build: no target point falls inside the grid"); rc = ICPX_EINVAL; break; }
        if (!counters[4] && !counters[5]) {
            if (!first_try) edt_adapt(ctx, counters[3], counters[7]);
            else if (getenv("ICPX_EDT_VERBOSE")) fprintf(stderr, "icpx edt: div %u first build with %d private levels fits\n", ctx->div, EDT_FIRST_CAP);
            break;
        }
        if (first_try) {   // deeper stacks than the private levels: the pool variant, with the report that shapes later builds
            ctx->edt_zy = ctx->edt_x = def;
            continue;
        }
        ctx->edt_overflowed = true;   // from now on the kernels with the pool / spill tiers, until a build fits
        // Some row's envelope did not fit the LDS levels and pool: give the passes a spill buffer and repeat
        // the build.
        if (with_spill || attempt == 2) { ctx->set_error("EDT build: stack overflow persisted with a spill buffer"); rc = ICPX_EHIP; break; }
        rc = icpx_reserve(ctx, ctx->edt_spill, edt_spill_bytes(ctx));
        if (rc != ICPX_OK) break;
    }
    if (ctx->edt_time_builds) {
        for (int k = 0; k < ICPX_EDT_NSTAGES; k++) {
            float ms = 0.f;
            if (rc == ICPX_OK) (void)hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]);
            ctx->edt_build_ms[k] = ms;
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    if (rc != ICPX_OK) return rc;
    ctx->edt_ready = true;
    return ICPX_OK;
}

extern "C" int icpx_edt_set_timing(icpx_ctx* ctx, int on) {
    if (!ctx) return ICPX_EINVAL;
    ctx->edt_time_builds = on != 0;
    return ICPX_OK;
}

extern "C" int icpx_edt_last_build(icpx_ctx* ctx, float stage_ms[ICPX_EDT_NSTAGES], int* attempts) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_last_build: no EDT has been built");
    if (stage_ms) for (int k = 0; k < ICPX_EDT_NSTAGES; k++) stage_ms[k] = ctx->edt_build_ms[k];
    if (attempts) *attempts = ctx->edt_build_attempts_made;
    return ICPX_OK;
}

extern "C" int icpx_edt_build_box(icpx_ctx* ctx, uint32_t div, const float min3[3], float cell_len) {
    if (!ctx) return ICPX_EINVAL;
    if (!min3 || !(cell_len > 0.f)) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_build_box: bad box");
    ICPX_TRY(edt_prepare(ctx, div));
    ctx->edt_round = false;
    ctx->edt_min[0] = min3[0]; ctx->edt_min[1] = min3[1]; ctx->edt_min[2] = min3[2];
    ctx->edt_cell = cell_len;
    return edt_build_attempts(ctx);
}

extern "C" int icpx_edt_build_round(icpx_ctx* ctx, uint32_t div, const double min3[3], double scale) {
    if (!ctx) return ICPX_EINVAL;
    if (!min3 || !(scale > 0.0)) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_build_round: bad box");
    ICPX_TRY(edt_prepare(ctx, div));
    ctx->edt_round = true;
    for (int k = 0; k < 3; k++) { ctx->edt_dmin[k] = min3[k]; ctx->edt_min[k] = (float)min3[k]; }
    ctx->edt_scale = scale;
    ctx->edt_cell = (float)(1.0 / scale);
    return edt_build_attempts(ctx);
}

extern "C" int icpx_edt_lookup_round(icpx_ctx* ctx, const double* xyz, size_t n, int where, float* dist) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready || !ctx->edt_round) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_lookup_round: no grid built with icpx_edt_build_round");
    if (n == 0) return ICPX_OK;
    if (!xyz || !dist) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_edt_lookup_round: NULL buffer");
    ICPX_TRY(icpx_bind(ctx));
    const double* dq = xyz; float* dout = dist;
    if (where == ICPX_HOST) {
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * 3 * sizeof(double)));
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_d2, n * sizeof(float)));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, xyz, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        dq = ctx->scratch_q.as<double>(); dout = ctx->scratch_d2.as<float>();
    }
    hipLaunchKernelGGL(k_edt_lookup_round, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->edt_sq.as<int>(), (int)ctx->div, (int)ctx->edt_zs,
                       ctx->edt_dmin[0], ctx->edt_dmin[1], ctx->edt_dmin[2], ctx->edt_scale, dq, n, dout);
    ICPX_TRY(icpx_launch_check(ctx, "k_edt_lookup_round"));
    if (where == ICPX_HOST)
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(dist, dout, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    return icpx_edt_run(ctx, false);
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
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(counters, edt_counters(ctx), sizeof counters, hipMemcpyDeviceToHost, ctx->stream));
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
    // the caller gets div^3 values, rows back to back; the grid's own rows are edt_zs values apart
    const hipMemcpyKind kind = where == ICPX_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (ctx->edt_zs == ctx->div) {
        const size_t bytes = (size_t)ctx->div * ctx->div * ctx->div * sizeof(int);
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(sq_out, ctx->edt_sq.p, bytes, kind, ctx->stream));
    } else {
        ICPX_CHECK_HIP(ctx, hipMemcpy2DAsync(sq_out, (size_t)ctx->div * sizeof(int), ctx->edt_sq.p, (size_t)ctx->edt_zs * sizeof(int),
                                             (size_t)ctx->div * sizeof(int), (size_t)ctx->div * ctx->div, kind, ctx->stream));
    }
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

extern "C" int icpx_edt_grid_ptr(icpx_ctx* ctx, const int32_t** sq_dev) {
    if (!ctx || !sq_dev) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_grid_ptr: no EDT has been built");
    *sq_dev = ctx->edt_sq.as<int32_t>();
    return ICPX_OK;
}

extern "C" int icpx_edt_grid_stride(icpx_ctx* ctx, uint32_t* stride) {
    if (!ctx || !stride) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_grid_stride: no EDT has been built");
    *stride = ctx->edt_zs;
    return ICPX_OK;
}

extern "C" int icpx_edt_lookup(icpx_ctx* ctx, const float* xyz, size_t n, int where, float* dist) {
    if (!ctx) return ICPX_EINVAL;
    if (!ctx->edt_ready) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_lookup: no EDT has been built");
    if (ctx->edt_round) ICPX_FAIL(ctx, ICPX_ESTATE, "icpx_edt_lookup: the grid was built with icpx_edt_build_round (use icpx_edt_lookup_round)");
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
    hipLaunchKernelGGL(k_edt_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->edt_sq.as<int>(), (int)ctx->div, (int)ctx->edt_zs,
                       ctx->edt_min[0], ctx->edt_min[1], ctx->edt_min[2], ctx->edt_cell, dq, n, dout);
    ICPX_TRY(icpx_launch_check(ctx, "k_edt_lookup"));
    if (where == ICPX_HOST)
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(dist, dout, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}
