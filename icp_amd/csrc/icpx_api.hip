// Context management, staging and the small shared kernels of the icpx C ABI.
#include <cstring>
#include <cstdlib>
#include <cfloat>
#include <new>

#include "icpx_ctx.h"

#include <rocprim/device/device_radix_sort.hpp>

static thread_local std::string g_create_error = "";

int icpx_bind(icpx_ctx* ctx) {
    ICPX_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    return ICPX_OK;
}

int icpx_reserve(icpx_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return ICPX_OK;
    if (b.p) {
        ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ICPX_CHECK_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        ctx->set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return ICPX_ENOMEM;
    }
    b.cap = want;
    return ICPX_OK;
}

int icpx_launch_check(icpx_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return ICPX_EHIP;
    }
    return ICPX_OK;
}

extern "C" int icpx_create(int device, icpx_ctx** out) {
    if (!out) { g_create_error = "icpx_create: out is NULL"; return ICPX_EINVAL; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("icpx_create: no HIP device available (") + hipGetErrorString(e) + ")";
        return ICPX_EHIP;
    }
    if (device < 0 || device >= ndev) {
        g_create_error = "icpx_create: device index out of range";
        return ICPX_EINVAL;
    }
    icpx_ctx* ctx = new (std::nothrow) icpx_ctx;
    if (!ctx) { g_create_error = "icpx_create: out of host memory"; return ICPX_ENOMEM; }
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("icpx_create: ") + hipGetErrorString(e);
        delete ctx;
        return ICPX_EHIP;
    }
    ctx->stream = ctx->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    // Tuning knobs of the EDT envelope passes: site groups per row block, LDS stack levels per lane and
    // group (a workgroup may use 64 KiB of LDS).
    // ICPX_EDT_ZY / ICPX_EDT_X = "groups,levels,chunks"; accepted only if the workgroup's LDS stays within 64 KiB.
    for (int k = 0; k < 2; k++) {
        const char* v = getenv(k ? "ICPX_EDT_X" : "ICPX_EDT_ZY");
        int g = 0, c = 0, n = 0;
        if (v && sscanf(v, "%d,%d,%d", &g, &c, &n) == 3 && g >= 1 && g <= 4 && c >= 2 && n >= 0 && n <= 254 &&
            g * 384 + 768 + g * (c + 1) * 384 + n * 48 <= 65536)
            { (k ? ctx->edt_x : ctx->edt_zy) = EnvTune{g, c, n}; ctx->edt_tune_fixed = true; }
    }
    // ICPX_ICP_TILES = "walk_all,ball_min": source clouds with fewer 64-point tiles than walk_all are walked point by point
    // without the tile kernel, clouds with at least ball_min tiles also use the ball kernel (defaults in icpx_ctx.h; the
    // tests set 0,0 to run small clouds through every kernel)
    {
        const char* v = getenv("ICPX_ICP_TILES");
        long long a = 0, b2 = 0;
        if (v && sscanf(v, "%lld,%lld", &a, &b2) == 2 && a >= 0 && b2 >= 0) { ctx->icp_walk_all_tiles = (size_t)a; ctx->icp_ball_min_tiles = (size_t)b2; }
    }
    { const char* v = getenv("ICPX_CAND"); long long a = 0; if (v && sscanf(v, "%lld", &a) == 1 && a >= 0) { ctx->cand_enable = a != 0; if (a > 1) ctx->cand_max_entries = (size_t)a; } }
    { const char* v = getenv("ICPX_SEARCH_FEW"); int x = 0; if (v && sscanf(v, "%d", &x) == 1 && x >= 0) ctx->search_few = x; }
    { const char* v = getenv("ICPX_SEARCH_STEPS"); int x = 0; if (v && sscanf(v, "%d", &x) == 1 && x >= 1) ctx->search_max_steps = x; }
    { const char* v = getenv("ICPX_SEARCH_LOT"); int x = 0, y = 0; if (v && sscanf(v, "%d,%d", &x, &y) == 2 && x >= 1 && y >= x) { ctx->search_lot = x; ctx->search_lot_max = y; } }
    { const char* v = getenv("ICPX_WALK_WIDE"); long long a = 0; if (v && sscanf(v, "%lld", &a) == 1 && a >= 0) ctx->icp_walk_wide_tiles = (size_t)a; }
    { const char* v = getenv("ICPX_BALL_FEW"); long long a = 0; if (v && sscanf(v, "%lld", &a) == 1 && a >= 0) ctx->icp_ball_few = (unsigned)a; }
    { const char* v = getenv("ICPX_BALL_MANY"); long long a = 0; if (v && sscanf(v, "%lld", &a) == 1 && a >= 0) ctx->icp_ball_many = (unsigned)a; }
    // ICPX_BNB_GROUP_MIN: bound batches of at least this many items go through the kernel that takes eight items per
    // workgroup (the tests set 0 and a huge value to compare the two kernels)
    {
        const char* v = getenv("ICPX_BNB_GROUP_MIN");
        long long a = 0;
        if (v && sscanf(v, "%lld", &a) == 1 && a >= 0) ctx->bnb_group_min = (size_t)a;
        v = getenv("ICPX_BNB_BRICKS");
        if (v && v[0] >= '0' && v[0] <= '2' && !v[1]) ctx->bnb_bricks = v[0] - '0';
    }
    if (icpx_bnb_scratch_init(ctx) != ICPX_OK) {
        g_create_error = "icpx_create: " + ctx->err;
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return ICPX_EHIP;
    }
    *out = ctx;
    return ICPX_OK;
}

extern "C" void icpx_destroy(icpx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->tgt_xyz, &ctx->cell_start, &ctx->cell_count, &ctx->pyr_masks, &ctx->pyr_tab, &ctx->sorted_pts, &ctx->sort_keys_a, &ctx->sort_keys_b,
                      &ctx->sort_vals_a, &ctx->sort_vals_b, &ctx->sort_tmp, &ctx->cand_tab, &ctx->cand_tmp, &ctx->src_pts, &ctx->nn_seed, &ctx->nn_pair, &ctx->icp_loop, &ctx->walk_queue, &ctx->tile_counters, &ctx->partials, &ctx->sums,
                      &ctx->scratch_q, &ctx->scratch_idx, &ctx->scratch_d2, &ctx->scratch_f, &ctx->edt_bits,
                      &ctx->edt_rowlist, &ctx->edt_rowcnt, &ctx->edt_planelist, &ctx->edt_counters, &ctx->trim_keys_a, &ctx->trim_keys_b, &ctx->trim_vals_a, &ctx->trim_vals_b, &ctx->trim_p, &ctx->trim_idx, &ctx->edt_spill, &ctx->edt_plane,
                      &ctx->edt_sq, &ctx->edt_bricks, &ctx->edt_fbricks, &ctx->data_pts, &ctx->maxrot, &ctx->bnb_items, &ctx->bnb_partials,
                      &ctx->bnb_mindis};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (DevBuf& b : ctx->search_buf)
        if (b.p) (void)hipFree(b.p);
    if (ctx->bnb_pin) (void)hipHostFree(ctx->bnb_pin);
    for (BnbSlot& sl : ctx->bnb_slot) {
        for (DevBuf* b : {&sl.parents, &sl.items, &sl.partials})
            if (b->p) (void)hipFree(b->p);
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.done) (void)hipEventDestroy(sl.done);

    }
    if (ctx->pin_sums) (void)hipHostFree(ctx->pin_sums);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char* icpx_last_error(const icpx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int icpx_set_stream(icpx_ctx* ctx, void* hip_stream) {
    if (!ctx) return ICPX_EINVAL;
    ICPX_TRY(icpx_bind(ctx));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;   // NULL = the default stream
    return ICPX_OK;
}

extern "C" int icpx_reset_stream(icpx_ctx* ctx) {
    if (!ctx) return ICPX_EINVAL;
    ICPX_TRY(icpx_bind(ctx));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return ICPX_OK;
}

extern "C" int icpx_sync(icpx_ctx* ctx) {
    if (!ctx) return ICPX_EINVAL;
    ICPX_TRY(icpx_bind(ctx));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}

extern "C" int icpx_device(const icpx_ctx* ctx) { return ctx ? ctx->device : -1; }

// ---- staging ---------------------------------------------------------------------------------------

__global__ void k_pack_points4(const char* __restrict__ src, size_t n, size_t stride, float4* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(src + i * stride);
    dst[i] = make_float4(p[0], p[1], p[2], __int_as_float((int)i));
}

int icpx_stage_points4(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where, float4* dst) {
    const char* dsrc = reinterpret_cast<const char*>(xyz);
    if (where == ICPX_HOST) {
        const size_t bytes = (n - 1) * stride_bytes + 12;
        ICPX_TRY(icpx_reserve(ctx, ctx->scratch_f, bytes));
        ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_f.p, xyz, bytes, hipMemcpyHostToDevice, ctx->stream));
        dsrc = ctx->scratch_f.as<char>();
    }
    const int threads = 256;
    const unsigned blocks = (unsigned)((n + threads - 1) / threads);
    hipLaunchKernelGGL(k_pack_points4, dim3(blocks), dim3(threads), 0, ctx->stream, dsrc, n, stride_bytes, dst);
    return icpx_launch_check(ctx, "k_pack_points4");
}

// Per-block min / max of float4 points -> partial[block] = {min xyz, max xyz, 1 if a coordinate is NaN or
// infinite (such a point would be filed under an arbitrary cell and poison every distance it meets)}.
__global__ __launch_bounds__(256) void k_bbox_partial(const float4* __restrict__ pts, size_t n, float* __restrict__ partial) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 p = pts[i];
        bad |= !(fabsf(p.x) <= FLT_MAX && fabsf(p.y) <= FLT_MAX && fabsf(p.z) <= FLT_MAX);
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
    __shared__ float sm[6][4];
    for (int c = 0; c < 3; c++) {
        for (int o = 32; o > 0; o >>= 1) {
            mn[c] = fminf(mn[c], __shfl_xor(mn[c], o));
            mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o));
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) for (int c = 0; c < 3; c++) { sm[c][wave] = mn[c]; sm[3 + c][wave] = mx[c]; }
    __syncthreads();
    const int any_bad = __syncthreads_or(bad);
    if (threadIdx.x < 6) {
        float v = sm[threadIdx.x][0];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, sm[threadIdx.x][w]) : fmaxf(v, sm[threadIdx.x][w]);
        partial[blockIdx.x * 7 + threadIdx.x] = v;
    }
    if (threadIdx.x == 6) partial[blockIdx.x * 7 + 6] = any_bad ? 1.f : 0.f;
}

static int device_bbox4(icpx_ctx* ctx, const float4* pts, size_t n, float mn[3], float mx[3]) {
    const int blocks = (int)std::min<size_t>(1024, (n + 255) / 256);
    ICPX_TRY(icpx_reserve(ctx, ctx->partials, sizeof(float) * 7 * 1024));
    hipLaunchKernelGGL(k_bbox_partial, dim3(blocks), dim3(256), 0, ctx->stream, pts, n, ctx->partials.as<float>());
    ICPX_TRY(icpx_launch_check(ctx, "k_bbox_partial"));
    std::vector<float> h(7 * blocks);
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(h.data(), ctx->partials.p, sizeof(float) * 7 * blocks, hipMemcpyDeviceToHost, ctx->stream));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < 3; c++) { mn[c] = h[c]; mx[c] = h[3 + c]; }
    bool bad = h[6] != 0.f;
    for (int b = 1; b < blocks; b++) {
        for (int c = 0; c < 3; c++) { mn[c] = std::min(mn[c], h[7 * b + c]); mx[c] = std::max(mx[c], h[7 * b + 3 + c]); }
        bad = bad || h[7 * b + 6] != 0.f;
    }
    if (bad) ICPX_FAIL(ctx, ICPX_EINVAL, "the cloud holds NaN or infinite coordinates (the reference's loader drops such points, src/main.cpp:35-51)");
    return ICPX_OK;
}

__device__ __forceinline__ unsigned spread10(unsigned v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void k_morton_keys(const float4* __restrict__ pts, size_t n, float3 mn, float3 scale, unsigned* __restrict__ keys,
                              unsigned* __restrict__ vals) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    unsigned ix = (unsigned)fminf(fmaxf((p.x - mn.x) * scale.x, 0.f), 1023.f);
    unsigned iy = (unsigned)fminf(fmaxf((p.y - mn.y) * scale.y, 0.f), 1023.f);
    unsigned iz = (unsigned)fminf(fmaxf((p.z - mn.z) * scale.z, 0.f), 1023.f);
    keys[i] = spread10(ix) | (spread10(iy) << 1) | (spread10(iz) << 2);
    vals[i] = (unsigned)i;
}

__global__ void k_gather4(const float4* __restrict__ src, const unsigned* __restrict__ order, size_t n, float4* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[order[i]];
}

// Sorts (keys_a, vals_a) -> (keys_b, vals_b), stable, on the context stream.
int icpx_sort_pairs(icpx_ctx* ctx, size_t n, unsigned end_bit) {
    size_t tmp = 0;
    unsigned* ka = ctx->sort_keys_a.as<unsigned>(); unsigned* kb = ctx->sort_keys_b.as<unsigned>();
    unsigned* va = ctx->sort_vals_a.as<unsigned>(); unsigned* vb = ctx->sort_vals_b.as<unsigned>();
    ICPX_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp, ka, kb, va, vb, n, 0, end_bit, ctx->stream));
    ICPX_TRY(icpx_reserve(ctx, ctx->sort_tmp, tmp));
    ICPX_CHECK_HIP(ctx, rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp, ka, kb, va, vb, n, 0, end_bit, ctx->stream));
    return ICPX_OK;
}

// Stable sort of (64-bit keys, 32-bit values), caller-provided double buffers, on the context stream.
int icpx_sort_pairs64(icpx_ctx* ctx, size_t n, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned* vals_in,
                      unsigned* vals_out) {
    size_t tmp = 0;
    ICPX_CHECK_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp, keys_in, keys_out, vals_in, vals_out, n, 0, 64, ctx->stream));
    ICPX_TRY(icpx_reserve(ctx, ctx->sort_tmp, tmp));
    ICPX_CHECK_HIP(ctx, rocprim::radix_sort_pairs(ctx->sort_tmp.p, tmp, keys_in, keys_out, vals_in, vals_out, n, 0, 64, ctx->stream));
    return ICPX_OK;
}

int icpx_sort_reserve(icpx_ctx* ctx, size_t n) {
    ICPX_TRY(icpx_reserve(ctx, ctx->sort_keys_a, n * 4));
    ICPX_TRY(icpx_reserve(ctx, ctx->sort_keys_b, n * 4));
    ICPX_TRY(icpx_reserve(ctx, ctx->sort_vals_a, n * 4));
    ICPX_TRY(icpx_reserve(ctx, ctx->sort_vals_b, n * 4));
    return ICPX_OK;
}

int icpx_morton_sort4(icpx_ctx* ctx, float4* pts, size_t n) {
    float mn[3], mx[3];
    ICPX_TRY(device_bbox4(ctx, pts, n, mn, mx));
    float3 lo = make_float3(mn[0], mn[1], mn[2]);
    float3 sc;
    sc.x = mx[0] > mn[0] ? 1023.f / (mx[0] - mn[0]) : 0.f;
    sc.y = mx[1] > mn[1] ? 1023.f / (mx[1] - mn[1]) : 0.f;
    sc.z = mx[2] > mn[2] ? 1023.f / (mx[2] - mn[2]) : 0.f;
    ICPX_TRY(icpx_sort_reserve(ctx, n));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_morton_keys, dim3(blocks), dim3(256), 0, ctx->stream, pts, n, lo, sc, ctx->sort_keys_a.as<unsigned>(),
                       ctx->sort_vals_a.as<unsigned>());
    ICPX_TRY(icpx_launch_check(ctx, "k_morton_keys"));
    ICPX_TRY(icpx_sort_pairs(ctx, n, 30));
    // gather through a temporary copy
    ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * sizeof(float4)));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, pts, n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather4, dim3(blocks), dim3(256), 0, ctx->stream, ctx->scratch_q.as<float4>(), ctx->sort_vals_b.as<unsigned>(), n, pts);
    return icpx_launch_check(ctx, "k_gather4");
}

// ---- compact tiles for the ICP pass ------------------------------------------------------------------
// The ICP pass works on tiles of 64 consecutive source points and stages, per tile, the target points around all 64
// queries: what a tile costs grows with the square of its extent. 64 consecutive points of a Morton order are not a
// compact group (the curve fills space, the points lie on a surface, and the curve jumps: on a million surface points
// the distance from a tile's centre to its farthest point averages 4.8 search cells where 64 points cover a patch of
// 2.2). So every block of CL_B Morton-consecutive points is cut into tiles by median splits, like the leaves of a
// balanced k-d tree: the block is sorted along the longest axis of its bounding box and cut in the middle, each half
// again along its own longest axis, down to runs of 64 (mean extent 2.9 cells, mean square 13 against 41). One
// workgroup per block; keys and the permutation live in LDS, the sorts of all segments of a level run as one bitonic
// network. Blocks keep their Morton order, so neighbouring tiles still lie in neighbouring places.
constexpr int CL_B = 8192, CL_T = 1024;

__global__ __launch_bounds__(CL_T) void k_tile_cluster(const float4* __restrict__ in, size_t n, float4* __restrict__ out) {
    __shared__ float s_key[CL_B];
    __shared__ unsigned short s_perm[CL_B];
    __shared__ float s_box[CL_B / 64][6];
    __shared__ unsigned char s_axis[CL_B / 128];
    const size_t base = (size_t)blockIdx.x * CL_B;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cnt = (int)(n - base < (size_t)CL_B ? n - base : (size_t)CL_B);   // positions beyond hold no point: they sort last at every level
    for (int i = t; i < CL_B; i += CL_T) s_perm[i] = (unsigned short)i;
    __syncthreads();
    for (int seg = CL_B; seg > 64; seg >>= 1) {
        // bounding boxes of the runs of 64 positions (a wave each), then of the segments
        for (int c = wave; c < CL_B / 64; c += CL_T / 64) {
            const int at = s_perm[c * 64 + lane];
            const bool real = at < cnt;
            const float4 p = real ? in[base + at] : make_float4(0.f, 0.f, 0.f, 0.f);
            float lo0 = real ? p.x : 3e38f, lo1 = real ? p.y : 3e38f, lo2 = real ? p.z : 3e38f;
            float hi0 = real ? p.x : -3e38f, hi1 = real ? p.y : -3e38f, hi2 = real ? p.z : -3e38f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo0 = fminf(lo0, __shfl_xor(lo0, o)); lo1 = fminf(lo1, __shfl_xor(lo1, o)); lo2 = fminf(lo2, __shfl_xor(lo2, o));
                hi0 = fmaxf(hi0, __shfl_xor(hi0, o)); hi1 = fmaxf(hi1, __shfl_xor(hi1, o)); hi2 = fmaxf(hi2, __shfl_xor(hi2, o));
            }
            if (lane == 0) { s_box[c][0] = lo0; s_box[c][1] = lo1; s_box[c][2] = lo2; s_box[c][3] = hi0; s_box[c][4] = hi1; s_box[c][5] = hi2; }
        }
        __syncthreads();
        const int nseg = CL_B / seg, per = seg / 64;
        if (t < nseg) {
            float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
            for (int c = t * per; c < (t + 1) * per; c++)
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], s_box[c][a]); hi[a] = fmaxf(hi[a], s_box[c][3 + a]); }
            const float e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
            s_axis[t] = (unsigned char)(e0 >= e1 && e0 >= e2 ? 0 : e1 >= e2 ? 1 : 2);
        }
        __syncthreads();
        for (int i = t; i < CL_B; i += CL_T) {
            const int at = s_perm[i];
            float key = 3e38f;
            if (at < cnt) { const float4 p = in[base + at]; const int a = s_axis[i / seg]; key = a == 0 ? p.x : a == 1 ? p.y : p.z; }
            s_key[i] = key;
        }
        __syncthreads();
        // every aligned segment sorted ascending by key: one bitonic network over the block, whose last merge (k == seg) runs
        // in the same direction in every segment
        for (int k = 2; k <= seg; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int q = t; q < CL_B / 2; q += CL_T) {
                    const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1)), p2 = i | j;
                    const bool asc = k == seg || (i & k) == 0;
                    const float a = s_key[i], b = s_key[p2];
                    if ((a > b) == asc && a != b) {
                        s_key[i] = b; s_key[p2] = a;
                        const unsigned short u = s_perm[i]; s_perm[i] = s_perm[p2]; s_perm[p2] = u;
                    }
                }
                __syncthreads();
            }
    }
    for (int i = t; i < cnt; i += CL_T) out[base + i] = in[base + s_perm[i]];
}

// Regroups spatially sorted points (icpx_morton_sort4) into compact tiles of 64; pts is rewritten in place.
int icpx_cluster_tiles4(icpx_ctx* ctx, float4* pts, size_t n) {
    ICPX_TRY(icpx_reserve(ctx, ctx->scratch_q, n * sizeof(float4)));
    ICPX_CHECK_HIP(ctx, hipMemcpyAsync(ctx->scratch_q.p, pts, n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_tile_cluster, dim3((unsigned)((n + CL_B - 1) / CL_B)), dim3(CL_T), 0, ctx->stream, ctx->scratch_q.as<float4>(), n, pts);
    return icpx_launch_check(ctx, "k_tile_cluster");
}

// ---- target ------------------------------------------------------------------------------------------

__global__ void k_unpack3(const float4* __restrict__ src, size_t n, float* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = src[i];
    dst[3 * i] = p.x; dst[3 * i + 1] = p.y; dst[3 * i + 2] = p.z;
}

extern "C" int icpx_set_target(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where) {
    if (!ctx) return ICPX_EINVAL;
    if (!xyz || n == 0) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_set_target: empty target cloud");
    if (stride_bytes < 12 || stride_bytes % 4) ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_set_target: stride must be a multiple of 4 and >= 12");
    if (n > 0x7fffffffu) ICPX_FAIL(ctx, ICPX_ELIMIT, "icpx_set_target: more than 2^31-1 points");
    ICPX_TRY(icpx_bind(ctx));
    ctx->grid_ready = false;
    ctx->edt_ready = false;
    ICPX_TRY(icpx_reserve(ctx, ctx->sorted_pts, n * sizeof(float4)));
    ICPX_TRY(icpx_reserve(ctx, ctx->tgt_xyz, n * 3 * sizeof(float)));
    float4* p4 = ctx->sorted_pts.as<float4>();
    ICPX_TRY(icpx_stage_points4(ctx, xyz, n, stride_bytes, where, p4));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_unpack3, dim3(blocks), dim3(256), 0, ctx->stream, p4, n, ctx->tgt_xyz.as<float>());
    ICPX_TRY(icpx_launch_check(ctx, "k_unpack3"));
    ICPX_TRY(device_bbox4(ctx, p4, n, ctx->bbmin, ctx->bbmax));
    for (int c = 0; c < 3; c++)
        if (!(ctx->bbmin[c] == ctx->bbmin[c]) || !(ctx->bbmax[c] - ctx->bbmin[c] < FLT_MAX))
            ICPX_FAIL(ctx, ICPX_EINVAL, "icpx_set_target: non-finite coordinates in the target cloud");
    ctx->nt = n;
    // the pairs remembered from passes against the previous target say nothing about this one
    if (ctx->nn_seed.p && ctx->ns) ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->nn_seed.p, 0xff, ctx->ns * sizeof(int), ctx->stream));
    if (ctx->nn_pair.p && ctx->ns) ICPX_CHECK_HIP(ctx, hipMemsetAsync(ctx->nn_pair.p, 0xff, ctx->ns * sizeof(float4), ctx->stream));
    ICPX_TRY(icpx_grid_build(ctx));
    ICPX_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ICPX_OK;
}
