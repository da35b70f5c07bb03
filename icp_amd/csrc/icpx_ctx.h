// Internal context of the icpx library (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/icpx.h"

// Grow-only device allocation.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct EnvTune { int groups, cap, chunks; };

// One bound batch in flight (icpx_bnb_submit / icpx_bnb_collect).
#define ICPX_BNB_SLOTS 2
struct BnbSlot {
    void* pin = nullptr;         // pinned host memory: the parent cubes going up, behind them the bounds coming down
    size_t pin_cap = 0;
    DevBuf parents, items, partials;
    hipEvent_t done = nullptr;   // recorded behind the batch's last command
    bool busy = false;           // submitted and not collected yet
    bool to_host = false;        // the bounds come back through `pin` (else they stay in the caller's device buffer)
    size_t nparents = 0;
    double* result = nullptr;    // where in `pin` the bounds arrive
};

struct icpx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    int num_cus = 256;

    // ---- target cloud --------------------------------------------------------------------------
    size_t nt = 0;
    DevBuf tgt_xyz;          // nt x 3 float, caller order
    float bbmin[3] = {0, 0, 0}, bbmax[3] = {0, 0, 0};   // float bounding box of the target

    // ---- uniform search grid over the target -----------------------------------------------------
    bool grid_ready = false;
    int gdim[3] = {0, 0, 0};
    float gmin[3] = {0, 0, 0};
    float gh = 0.f, ginv = 0.f;
    DevBuf cell_start;       // ncells + 1 int: position in sorted_pts of every cell's first point (exclusive prefix sum of the counts)
    DevBuf cell_count;       // ncells + 1 int: points per cell (input of that prefix sum)
    DevBuf pyr_masks;        // occupancy pyramid above the cells: one byte per node of every level >= 1, the mask of its children that hold points
    DevBuf pyr_tab;          // 11 unsigned: offset of every level inside pyr_masks
    int pyr_nlev = 0;        // levels including level 0 (the cells)
    int pyr_dim[11][3] = {};
    unsigned pyr_off[11] = {};
    DevBuf sorted_pts;       // nt x float4 {x, y, z, bits(original index)} sorted by cell, ties by index
    DevBuf sort_keys_a, sort_keys_b, sort_vals_a, sort_vals_b, sort_tmp;
    // candidate table over blocks of 2^cand_shift cells per axis: position in sorted_pts of a target point near the block's centre
    DevBuf cand_tab, cand_tmp;
    int cand_dim[3] = {0, 0, 0}, cand_shift = 1;
    bool cand_ready = false, cand_enable = true;   // ICPX_CAND=0 switches the table off
    size_t cand_max_entries = 3000000;             // ICPX_CAND=<entries>

    // ---- source cloud (ICP) ----------------------------------------------------------------------
    size_t ns = 0;
    DevBuf src_pts;          // ns x float4 {x, y, z, bits(original index)}, spatially sorted
    DevBuf icp_loop;         // icpx_icp_run: the loop's state (accumulated transform, running target mean), the sums of a pass, the per-iteration trace
    DevBuf nn_pair;          // ns float4, by sorted source position: the target point (x, y, z, index bits) the source point was paired with by the last ICP pass, index bits -1 = none
    DevBuf nn_seed;          // ns int, by sorted source position: position in sorted_pts of the target point the source point was paired with by the last ICP pass, -1 = none
    DevBuf tile_counters;    // eight 64-bit counters, 64 B apart, that hand out the tiles of an ICP pass; never reset
    unsigned long long tile_counter_base = 0;   // their common value before the next launch
    size_t icp_walk_wide_tiles = 65536;         // source clouds with at least this many tiles (4 M points) launch the walk kernel over the whole GPU (ICPX_WALK_WIDE)
    size_t icp_walk_all_tiles = 1024;           // source clouds with fewer tiles skip the tile kernel: every point is walked (ICPX_ICP_TILES)
    size_t icp_ball_min_tiles = 2048;           // source clouds with at least this many tiles use the ball kernel for far tiles
    unsigned icp_ball_few = 1024;               // with fewer far tiles than this the ball kernel hands their open lanes to k_icp_solo (ICPX_BALL_FEW)
    unsigned icp_ball_many = 8;                 // far tiles per wave of the ball kernel from which the large point / column caps apply (ICPX_BALL_MANY)
    DevBuf walk_queue;       // one ICP pass: the source points left to the pyramid walk, in 64 parts: a counter per part (16 words apart, zero between passes), then the parts' entries
    DevBuf partials;         // per-block partial sums
    DevBuf sums;             // ICPX_NSUMS doubles
    DevBuf scratch_q, scratch_idx, scratch_d2, scratch_f;   // staging for host-pointer calls
    // trimmed ICP step, all indexed by the caller's source index: squared distances as sort keys, iota
    // values, transformed points (3 doubles), correspondences
    DevBuf trim_keys_a, trim_keys_b, trim_vals_a, trim_vals_b, trim_p, trim_idx;

    // ---- EDT ---------------------------------------------------------------------------------------
    bool edt_ready = false;
    uint32_t div = 0;
    int edt_words = 0;       // 64-bit words per (x,y) bit row
    uint32_t edt_zs = 0;     // values between consecutive rows of edt_plane / edt_sq (z runs fastest): div, or div rounded up to 32 for large grids (see edt_prepare)
    float edt_min[3] = {0, 0, 0};
    float edt_cell = 0.f;
    bool edt_round = false;  // grid built with DT3D's convention: voxel = int((p - dmin) * scale + 0.5) in double
    double edt_dmin[3] = {0, 0, 0}, edt_scale = 0.0;
    // Envelope passes: site groups (= waves) per 64 rows, private LDS levels per lane and group, pool
    // chunks per workgroup. 4 x (28 + 1) x 64 x 6 bytes + 130 chunks x 48 bytes + header = 53 088 B: three
    // workgroups per CU, as with the 40 chunks used at first; the larger pool takes 4 to 12 % off pass x on
    // grids of 512^3 and more (bunny, 1023^3: 2.79 against 3.15 ms), whose deep stacks otherwise reach the spill tier.
    EnvTune edt_zy{4, 28, 130}, edt_x{4, 28, 130};
    bool edt_tune_fixed = false;   // shapes came from the environment: do not adapt them to the data
    bool edt_overflowed = false;   // a build needed levels beyond the private ones: use the kernels with pool / spill tiers
    uint32_t edt_shape_div = 0;    // resolution the shapes above were chosen at (0: none yet); another resolution starts afresh
    bool edt_time_builds = false;  // icpx_edt_set_timing: checked builds carry start / stop events per stage
    float edt_build_ms[ICPX_EDT_NSTAGES] = {0, 0, 0, 0};   // kernel times of the last checked build's final attempt, by stage
    int edt_build_attempts_made = 0;                        // attempts the last checked build needed
    DevBuf edt_bits;         // div*div*edt_words uint64 occupancy bits, then two sets of 8 uint64 counters (alternate builds)
    int edt_epoch = 0;       // counter set of the latest build
    DevBuf edt_rowlist;      // [x][chunk][k] 16-byte records of the non-empty rows of every plane (see k_edt_census)
    DevBuf edt_rowcnt;       // div int
    DevBuf edt_planelist;    // int32 ascending non-empty planes, padded with the last one to a multiple of 16 plus 16
    DevBuf edt_counters;     // scratch words of the search-grid build
    DevBuf edt_spill;        // NULL until a build overflowed the LDS stacks: [level - cap][group][row] 8-byte entries
    DevBuf edt_plane;        // div^3 int32: in-plane squared distances (only non-empty planes written)
    DevBuf edt_sq;           // div^3 int32: squared EDT, [x][y][z]
    DevBuf edt_bricks;       // the same values as uint16 in bricks of 4 x 4 x 4 voxels, for the bound kernel (made on first use after a build)
    DevBuf edt_fbricks;      // the distances (float) in the same brick order (bnb_bricks == 2)
    bool edt_bricks_ready = false, edt_fbricks_ready = false;

    // ---- Go-ICP ---------------------------------------------------------------------------------------
    size_t nd = 0;
    DevBuf data_pts;         // nd x float4 {x, y, z, bits(original index)}, spatially sorted
    DevBuf maxrot;           // ICPX_MAXROTLEVEL x nd float, sorted order
    DevBuf bnb_items, bnb_partials, bnb_mindis;
    // Batches of at least this many items are evaluated eight items per workgroup (k_bnb_eval8: half the instructions per
    // cube for the families of eight sibling cubes the search sends, an eighth of the workgroups); smaller ones one item
    // per workgroup, which fills the chip from 128 items on. Same bounds either way. ICPX_BNB_GROUP_MIN overrides.
    size_t bnb_group_min = 256;
    int bnb_bricks = 2;          // ICPX_BNB_BRICKS: 0 the bound kernel reads the [x][y][z] grid only, 1 the 16-bit bricks, 2 the float bricks
    double* pin_sums = nullptr;  // pinned host memory for the ICPX_NSUMS sums of an ICP pass: the reduction kernel writes them there itself
    double* bnb_scratch = nullptr;   // this GPU's address of the fixed partial-sum area of icpx_bnb_eval_noalloc (resolved at icpx_create)
    BnbSlot bnb_slot[ICPX_BNB_SLOTS];
    DevBuf search_buf[8];        // icpx_bnb_search_run: states, queues (cubes, bounds), lists, parent records, partial sums, list lengths, in / out records
    int search_few = 128;        // with at most this many searches running, the bound kernel keeps four points in flight per thread (ICPX_SEARCH_FEW)
    int search_max_steps = 4096; // steps after which icpx_bnb_search_run gives up on the searches still running (ICPX_SEARCH_STEPS lowers it)
    int search_lot = 4, search_lot_max = 16;   // steps enqueued between two looks at the list length: first lot, and the most (ICPX_SEARCH_LOT=first,most)
    void* bnb_pin = nullptr;     // pinned host staging of a bound batch: the items going up, then ub / lb coming down
    size_t bnb_pin_cap = 0;

    void set_error(const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
    }
};

#define ICPX_CHECK_HIP(ctx, call)                                                              \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? ICPX_ENOMEM : ICPX_EHIP;                        \
        }                                                                                      \
    } while (0)

#define ICPX_TRY(expr)                 \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != ICPX_OK) return rc_; \
    } while (0)

#define ICPX_FAIL(ctx, code, ...)       \
    do {                               \
        (ctx)->set_error(__VA_ARGS__); \
        return (code);                 \
    } while (0)

// icpx_api.hip
int icpx_reserve(icpx_ctx* ctx, DevBuf& b, size_t bytes);
int icpx_bind(icpx_ctx* ctx);   // hipSetDevice(ctx->device)
int icpx_launch_check(icpx_ctx* ctx, const char* what);
// Packs `n` points of 3 floats at `stride_bytes` (host or device) into a device float4 array
// {x, y, z, bits(i)} in caller order; dst must hold n float4.
int icpx_stage_points4(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where, float4* dst);
// Sorts float4 points (w = original index bits) by a 30-bit Morton code over their own bounding box.
int icpx_morton_sort4(icpx_ctx* ctx, float4* pts, size_t n);
// Cuts every block of 8192 consecutive points of such an order into compact runs of 64 (median splits), in place.
int icpx_cluster_tiles4(icpx_ctx* ctx, float4* pts, size_t n);

// icpx_nn.hip
int icpx_grid_build(icpx_ctx* ctx);
// icpx_bnb.hip
int icpx_bnb_scratch_init(icpx_ctx* ctx);
// icpx_edt.hip
int icpx_edt_run(icpx_ctx* ctx, bool report);
