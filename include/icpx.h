/* icpx.h — C ABI of the MI355X (gfx950) ICP / Go-ICP correspondence engine.
 *
 * One opaque context per GPU. Every function returns ICPX_OK (0) or a negative ICPX_E* code; the text
 * of the last failure is available from icpx_last_error(). Nothing throws, nothing calls exit().
 * The caller owns every buffer it passes in; the context owns its device buffers. A context is used
 * from one host thread at a time. Calls are synchronous (they return after the work on the context's
 * stream has completed) except those suffixed _async, which only enqueue on the context's stream and
 * are completed by icpx_sync().
 *
 * Pointer arguments that carry bulk data come with a `where` argument: ICPX_HOST (pageable or pinned
 * host memory, copied by the library) or ICPX_DEVICE (memory of the context's GPU, used in place).
 * Small fixed-size arguments (matrices, boxes, cube lists) are always host memory unless stated.
 *
 * The reference (Hcque/ICP) has no FFI: these entry points replace member functions of its C++
 * classes; each declaration cites the member it stands in for (paths relative to the reference root).
 */
#ifndef ICPX_H
#define ICPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct icpx_ctx icpx_ctx;

enum { ICPX_HOST = 0, ICPX_DEVICE = 1 };

enum {
    ICPX_OK = 0,
    ICPX_EINVAL = -1,   /* bad argument (null pointer, zero size, out-of-range value) */
    ICPX_ESTATE = -2,   /* call order: a prerequisite (target / source / EDT / data) is missing */
    ICPX_EHIP = -3,     /* a HIP runtime call failed; the text carries hipGetErrorString */
    ICPX_ENOMEM = -4,   /* device allocation failed */
    ICPX_ELIMIT = -5    /* a documented size limit was exceeded */
};

/* Number of doubles icpx_icp_step writes: sum p (3), sum q (3), sum p q^T (9, row-major p_r*q_c),
 * sum |p-q|^2 (1). */
#define ICPX_NSUMS 16
/* Rotation-uncertainty levels of GoIcp::Init (src/GoIcp.cpp:20, MAXROTLEVEL). */
#define ICPX_MAXROTLEVEL 20
/* Largest EDT resolution per axis (stack entries pack indices in 10 bits). */
#define ICPX_MAX_DIV 1023

/* ---- context ------------------------------------------------------------------------------------ */

int icpx_create(int device, icpx_ctx** out);
void icpx_destroy(icpx_ctx* ctx);
/* Text of the most recent error on this context (or of the last failed icpx_create when ctx is
 * NULL). Never NULL. */
const char* icpx_last_error(const icpx_ctx* ctx);
/* Use an existing hipStream_t (passed as void*) for all work of this context; NULL means HIP's default
 * (null) stream, which is what torch.cuda.current_stream() is unless the caller switched streams.
 * icpx_reset_stream goes back to the stream the context created for itself. Both wait for the work
 * already enqueued on the previous stream. */
int icpx_set_stream(icpx_ctx* ctx, void* hip_stream);
int icpx_reset_stream(icpx_ctx* ctx);
int icpx_sync(icpx_ctx* ctx);
int icpx_device(const icpx_ctx* ctx);

/* ---- target cloud + nearest neighbour -------------------------------------------------------------- */

/* Replaces KDTree::KDTree(cloud, 3) + KDTree::build (src/kdtree.cpp:58-89): takes the target (model)
 * cloud, n points of 3 floats at byte stride `stride_bytes` (12 for packed xyz, 16 for the PCL
 * point layout), and builds the uniform search grid on the GPU. Also records the float bounding box
 * that LDT uses (src/testLDT.cpp:50). */
int icpx_set_target(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where);

/* Replaces the query loop of SingleThreadIcp::kdtree_search over KDTree::get_closest
 * (src/SingleThreadIcp.cpp:194-209, src/kdtree.cpp:107-150): for each of n double-precision queries
 * (row-major n x 3) the index of the nearest target point and the squared distance
 * ((dx*dx + dy*dy) + dz*dz in IEEE double without fused multiply-add, src/kdtree.cpp:131-133).
 * Ties resolve to the lowest target index. idx and d2 may each be NULL; they live where q lives. */
int icpx_nn(icpx_ctx* ctx, const double* q_xyz, size_t n, int where, int32_t* idx, double* d2);

/* Replaces KDTree::k_close(p, k, 1, Q) (src/kdtree.cpp:124-150): the k nearest target points of every query,
 * ascending by squared distance (the reference's priority queue pops them in descending order), equal
 * distances by ascending index. idx: n x k int32, d2: n x k doubles, either may be NULL; with fewer than k
 * target points the remaining slots get index -1 and an infinite distance. 1 <= k <= ICPX_MAX_KNN. */
#define ICPX_MAX_KNN 8
int icpx_knn(icpx_ctx* ctx, const double* q_xyz, size_t n, int where, int k, int32_t* idx, double* d2);

/* Replaces KDTree::Distance (src/kdtree.cpp:91-105): float queries, widened to double; returns
 * (float)sqrt(min squared distance). n x 3 floats in, n floats out. */
int icpx_nn_distance(icpx_ctx* ctx, const float* q_xyz, size_t n, int where, float* dist);

/* ---- ICP iteration: transform + NN + residual / cross-covariance reduction ------------------------ */

/* Replaces SingleThreadIcp::setSource (src/SingleThreadIcp.cpp:66-73): the source (data) cloud that
 * icpx_icp_step transforms. The library may keep it in a spatially sorted order internally;
 * per-point outputs are always returned in the caller's order. */
int icpx_set_source(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where);

/* One pass of the hot loop of SingleThreadIcp::registration (src/SingleThreadIcp.cpp:307-316):
 * p_i = T * [src_i; 1] in double (T = row-major 4x4, host memory, the ACCUMULATED transform
 * finMat), q_i = nearest target point of p_i (kdtree_search, :186-214), and the sums that
 * cal_covarance (:216-247) needs, and from which the residual that best_fit_transform logs after its update (:281-287)
 * follows without another pass over the points (icp::SingleThreadIcp::best_fit_transform has the formula):
 *   sums[0..2] = sum p, sums[3..5] = sum q, sums[6..14] = sum p q^T, sums[15] = sum |p - q|^2.
 * The caller forms the means, H = (sum p q^T - pbar (sum q)^T) / n, and runs the 3x3 SVD on the host.
 * sums: 16 doubles of host memory. idx_opt: NULL, or n int32 (memory kind `idx_where`) receiving the
 * correspondences in the caller's point order. The context remembers every source point's correspondence and
 * starts the next pass's search from it; that only prunes the search, the pairs are those of a search from scratch. */
int icpx_icp_step(icpx_ctx* ctx, const double T[16], double sums[ICPX_NSUMS], int32_t* idx_opt, int idx_where);

/* Same, but only enqueued: sums_dev is DEVICE memory (16 doubles) valid after icpx_sync() or after
 * any later work on the same stream — used when the sums are all-reduced across GPUs before the
 * host reads them. */
int icpx_icp_step_async(icpx_ctx* ctx, const double T[16], double* sums_dev);

/* The whole loop of SingleThreadIcp::registration (src/SingleThreadIcp.cpp:293-327: for every iteration kdtree_search,
 * cal_covarance, best_fit_transform, finMat = curMat * finMat) on the context's stream without the host in between:
 * the sums of a pass stay on the device, one thread forms the means (with the reference's carry-over of the previous
 * target mean, :229-231), H (/ (float)Nd, :244), the 3x3 SVD with the reflection fix (:256-266) and the update — the
 * arithmetic of icp::SingleThreadIcp's host code, compiled from the same source, so the result is the host loop's bit
 * for bit — and the next pass reads the accumulated transform from device memory. For one process holding the whole
 * source cloud (several ranks exchange their sums between the passes: icpx_icp_step_async).
 * T0: the starting transform (row-major 4x4, the caller's guess). mean_tar (in / out, 3 doubles): the target mean the
 * loop starts from (cal_mean of the target, src/SingleThreadIcp.cpp:53, :64) and the running value after the last
 * iteration. T_out: finMat after `iterations` iterations. trace: NULL, or 18 doubles per iteration: that iteration's
 * curMat (16, row-major), the residual after its update (sum |T p - q|^2 / Nd over the pairs it was fitted to, :281-288)
 * and before it (sum |p - q|^2 / Nd). All host memory; the call returns when the loop has finished. */
int icpx_icp_run(icpx_ctx* ctx, const double T0[16], int iterations, double mean_tar[3], double T_out[16], double* trace);

/* Trimmed form of the same pass, the correspondence step of ICP3D::Run with do_trim
 * (src/jly_icp3d.cpp:210-255): nearest target point of every transformed source point, then the sums of
 * icpx_icp_step over the `keep` pairs with the smallest squared distances only (the reference sorts all
 * pairs with qsort and uses the first num = (int)(n * (1 - trim_fraction))). Pairs with equal distance
 * at the cut are taken in the order of the caller's source index; the reference's order there is
 * whatever its qsort leaves. keep == n gives the sums of icpx_icp_step. */
int icpx_icp_step_trimmed(icpx_ctx* ctx, const double T[16], uint64_t keep, double sums[ICPX_NSUMS]);

/* ---- Euclidean distance transform --------------------------------------------------------------------- */

/* Replaces LDT::LDT(cloud, div, cores) (src/testLDT.cpp:40-212) for the current target: voxelisation
 * box by the reference's rule (:50-69), splat (:74-88), and the exact squared EDT that the three
 * passes (:91-206) compute. The parity object is the int32 squared grid before the final sqrt. */
int icpx_edt_build(icpx_ctx* ctx, uint32_t div);

/* Same with an explicit box (lower corner + cell length) instead of the reference's rule. Points whose
 * voxel falls outside the grid are dropped. */
int icpx_edt_build_box(icpx_ctx* ctx, uint32_t div, const float min3[3], float cell_len);

/* The same exact transform on a grid with DT3D's convention (src/jly_3ddt.cpp:792-798, :821-856): voxel index
 * int((p - min) * scale + 0.5) per axis in double. DT3D::Build derives min and scale from the cloud
 * (:731-764); the caller passes them (icp::DT3D in the host library applies that rule). The grid then
 * serves icpx_edt_lookup_round, which reproduces DT3D::Distance: rounded index; inside the grid
 * (float)(sqrt(squared cells) / scale); outside, the clamped voxel's value plus the overshoot in cells over
 * scale. DT3D itself fills its grid with an approximate propagation (DEuclidean); the values here are the
 * exact distances, so they differ from DT3D's by its approximation error. n x 3 doubles in, n floats out. */
int icpx_edt_build_round(icpx_ctx* ctx, uint32_t div, const double min3[3], double scale);
int icpx_edt_lookup_round(icpx_ctx* ctx, const double* q_xyz, size_t n, int where, float* dist);

/* Only the kernels of icpx_edt_build, enqueued on the stream (box and buffers from the previous
 * icpx_edt_build* call are reused): the timed region of the EDT benchmark. */
int icpx_edt_rebuild_async(icpx_ctx* ctx);

/* Runs the same kernels `reps` times back to back on the stream and waits. In every `sample_every`-th
 * build (the first one included) each launch carries its own start and stop HIP events; stage_ms
 * receives the kernel time of every stage summed over those builds, in milliseconds: splat, census,
 * pass zy (z scan + y envelope), pass x; *sampled receives their number. Events cost stream time, so
 * sample_every > 1 keeps the loop close to what icpx_edt_rebuild_async alone would take; the gaps between
 * kernels belong to no stage, the caller's clock around the call sees them. The reference times its
 * constructor with omp_get_wtime (src/testLDT.cpp:43, :209). */
#define ICPX_EDT_NSTAGES 4
int icpx_edt_rebuild_timed(icpx_ctx* ctx, int reps, int sample_every, float stage_ms[ICPX_EDT_NSTAGES], int* sampled);

/* The first build for a target is the one the reference's use pays for (one LDT per target, src/testLDT.cpp:40-212,
 * timed there with omp_get_wtime, :43 / :209). icpx_edt_set_timing(ctx, 1) makes every later icpx_edt_build* call
 * carry start / stop HIP events per stage; icpx_edt_last_build then returns the kernel milliseconds by stage of the
 * attempt that produced the grid, and the number of attempts the build needed (1: the first choice of stack storage
 * held; more: a pass reported an overflow and the kernels ran again with more storage). Either output may be NULL. */
int icpx_edt_set_timing(icpx_ctx* ctx, int on);
int icpx_edt_last_build(icpx_ctx* ctx, float stage_ms[ICPX_EDT_NSTAGES], int* attempts);

/* LDT's public fields (src/testLDT.cpp:31-35): div, _min, cellLen; plus the occupied-voxel count and
 * the number of points dropped by the splat. Any output pointer may be NULL. */
int icpx_edt_info(icpx_ctx* ctx, uint32_t* div, float min3[3], float* cell_len, uint64_t* occupied, uint64_t* dropped);

/* Copies the squared grid out: div^3 int32 in the reference's [x][y][z] order (z fastest). */
int icpx_edt_read(icpx_ctx* ctx, int32_t* sq_out, int where);

/* Device address of the squared grid as the library holds it, valid until the next build/destroy: int32, voxel
 * (x, y, z) at ((x * div + y) * stride + z), stride from icpx_edt_grid_stride. The stride is div for grids below 512
 * cells a side; from 512 on it is div rounded up to a multiple of 32, so that every row starts on a 128-byte line
 * (icpx_edt_read hands out the plain div^3 array either way). No counterpart in the reference (its LDT keeps the grid
 * in host memory, src/testLDT.cpp:31-35). */
int icpx_edt_grid_ptr(icpx_ctx* ctx, const int32_t** sq_dev);
int icpx_edt_grid_stride(icpx_ctx* ctx, uint32_t* stride);

/* Replaces LDT::Distance (src/testLDT.cpp:235-290): clamp into the box, truncate to a voxel,
 * sqrt(sq) * cellLen + overshoot. A coordinate equal to the upper clamp indexes voxel `div` in the
 * reference (out of range); here it is clamped to div-1. n x 3 floats in, n floats out. */
int icpx_edt_lookup(icpx_ctx* ctx, const float* xyz, size_t n, int where, float* dist);

/* ---- Go-ICP bound evaluation ---------------------------------------------------------------------------- */

/* Replaces the data-side state of GoIcp::GoIcp + GoIcp::Init (src/GoIcp.cpp:82-103, :459-489): uploads
 * the data cloud and builds maxRotDist[level][i] = (float)(2 sin(min(sqrt3 * sigma_l, pi) / 2) * |x_i|)
 * for ICPX_MAXROTLEVEL levels with the reference's truncated constants (:17-18). Requires the EDT. */
int icpx_goicp_set_data(icpx_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, int where);

/* Copies the maxRotDist table out (ICPX_MAXROTLEVEL x n floats, host memory, caller's point order). */
int icpx_goicp_read_maxrotdist(icpx_ctx* ctx, float* out);

/* One work item of a bound-evaluation batch: the loop body of GoIcp::innerBnB for one translation
 * child cube (src/GoIcp.cpp:384-431) under the rotation R (row-major 3x3 float, computed on the host
 * as at :243-261). level < 0: upper-bound form (maxRotDistL == NULL); 0 <= level < ICPX_MAXROTLEVEL:
 * lower-bound form subtracting maxRotDist[level]. */
typedef struct icpx_bnb_item {
    float R[9];
    float x, y, z, w; /* child cube: lower corner and side (TransNode, src/GoIcp.cpp:35-42) */
    int32_t level;
} icpx_bnb_item;

/* Evaluates nitems cubes in one launch. For every item: d_i = max(0, Distance(R x_i + centre) -
 * maxRotDist[level][i]) with the float arithmetic of src/GoIcp.cpp:268-271 and :388-411, then
 *   ub = sum d_i^2,   lb = sum max(0, d_i - (float)(sqrt3/2 * w))^2.
 * ub, lb: nitems doubles each (host), accumulated on the device in double (tree order).
 * mindis_opt: NULL, or nitems x n floats (host) receiving every d_i in the caller's point order, from
 * which the reference's sequential float sums (:416-431) can be reproduced bit for bit. */
int icpx_bnb_eval(icpx_ctx* ctx, const icpx_bnb_item* items, size_t nitems, double* ub, double* lb, float* mindis_opt);

/* Enqueue-only variant for several GPUs that share one batch: this context evaluates items first .. first + count - 1
 * of the nitems items and writes all of out_dev — DEVICE memory, 2 * nitems doubles laid out
 * [ub_0 .. ub_{n-1}, lb_0 .. lb_{n-1}] — with zeros in the entries of the other items, so that a sum over the ranks
 * (an all-reduce on the same stream) leaves every bound on every rank. first = 0, count = nitems evaluates the whole
 * batch. items is host memory (the whole batch) and is copied before the call returns (through one pinned staging area
 * per context: the call first waits for earlier work on the context's stream). The loop being shared is the child loop of
 * GoIcp::innerBnB (src/GoIcp.cpp:381-431). */
int icpx_bnb_eval_async(icpx_ctx* ctx, const icpx_bnb_item* items, size_t nitems, size_t first, size_t count, double* out_dev);

/* The search loop's form of the same evaluation: a record is a PARENT translation cube (x, y, z, w its corner and side,
 * R and level as above) and stands for its eight children in the order of GoIcp::innerBnB's child loop
 * (src/GoIcp.cpp:378-390: side w / 2, corner + (j & 1, j >> 1 & 1, j >> 2 & 1) * side, in float), so a batch goes up as an
 * eighth of the records; the bounds are those icpx_bnb_eval gives for the eight children, bit for bit. Two batches can
 * be in flight, one per slot (0 or 1): icpx_bnb_submit copies the records, enqueues the work on the context's stream
 * and returns; icpx_bnb_collect waits for that slot's batch alone and hands out ub / lb, 8 * nparents doubles each
 * (host), child j of parent p at 8 p + j. While one batch is evaluated the caller can digest the other's bounds.
 * This context evaluates parents first .. first + count - 1 of the batch (first = 0, count = nparents: all of it).
 * out_dev == NULL: the bounds come back to the host through icpx_bnb_collect. out_dev != NULL (DEVICE memory,
 * 16 * nparents doubles [ub..., lb...]): they are left there, zeros for the other parents' children, for a sum over
 * the ranks on the same stream as with icpx_bnb_eval_async; icpx_bnb_collect then only waits (ub / lb may be NULL).
 * Submitting to a slot whose batch was not collected waits for that batch and drops its bounds. */
int icpx_bnb_submit(icpx_ctx* ctx, int slot, const icpx_bnb_item* parents, size_t nparents, size_t first, size_t count, double* out_dev);
int icpx_bnb_collect(icpx_ctx* ctx, int slot, double* ub, double* lb);
/* For a batch that was submitted with a device buffer: enqueues, behind whatever the caller has put on the context's
 * stream since (the sum over the ranks), the copy of out_dev's 16 * nparents doubles to the host, so that
 * icpx_bnb_collect waits for that copy and hands out ub / lb as for a batch submitted with out_dev == NULL. */
int icpx_bnb_fetch(icpx_ctx* ctx, int slot, const double* out_dev);

/* Whole inner searches on the device: the loop of GoIcp::innerBnB (src/GoIcp.cpp:353-457) for n independent
 * (rotation, level) pairs at once, each from its own starting cube and best error so far. Per search: the priority queue
 * of translation cubes (lowest lower bound first, equal bounds in the order std::priority_queue<TransNode> gives them,
 * :361, :371), the eight children of the cube taken (:378-390), their bounds as icpx_bnb_eval gives them, the better
 * upper bound recorded with its cube (:432-437), children with lb >= best error dropped (:439), the search ended by
 * the first child whose lb is within sse_thresh of the best error (:446-450; float minus double, compared in double as
 * there) or by an empty queue (:367). The steps of all searches run as launches on the context's stream without the host in between; the call
 * returns when every search has ended.
 * found[i]: the best error (the starting value if no child improved on it), whether a cube gave it and that cube, the
 * cubes taken off the queue (8 bound evaluations each) and status: 0 ended as above; 1 the queue outgrew 1024 cubes;
 * 2 still running after 4096 steps. A search with status != 0 holds no result: run it by other means
 * (icpx_bnb_submit). steps_out (may be NULL): steps enqueued (one launch each). At most ICPX_SEARCH_MAX searches per
 * call (each holds 21 KB of device memory for its queue while the call runs). */
#define ICPX_SEARCH_MAX 65536
typedef struct icpx_bnb_search {
    float R[9];       /* rotation, row-major */
    int32_t level;    /* as icpx_bnb_item.level */
    float opt_error;  /* best error when the search starts (optErrorT = optError, :356) */
    float cube[4];    /* the starting cube: corner x, y, z and side (initNodeTrans, :359) */
} icpx_bnb_search;
typedef struct icpx_bnb_found {
    float opt_error;
    int32_t have_best;
    float best[4];
    int32_t cubes;
    int32_t status;
} icpx_bnb_found;
int icpx_bnb_search_run(icpx_ctx* ctx, const icpx_bnb_search* searches, size_t n, float sse_thresh, icpx_bnb_found* found, long long* steps_out);

/* Waits for the context's stream and copies `bytes` of device memory to host memory (results of the _async entry
 * points that the caller left in its own device buffers). No counterpart in the reference. */
int icpx_download(icpx_ctx* ctx, const void* dev, void* host, size_t bytes);
/* The other way: `bytes` from host memory to a device address, enqueued on the context's stream (through the context's
 * pinned staging area; the host buffer may be reused as soon as the call returns). The host classes use the pair for
 * the small result buffers they hand to a caller's device collective (GoIcp's per-step sum over the ranks; the reference
 * is a single process and has no counterpart, src/GoIcp.cpp:203-344 is the loop whose cubes are dealt out). */
int icpx_upload(icpx_ctx* ctx, void* dev, const void* host, size_t bytes);

/* No counterpart in the reference. For callers that cannot allocate device memory while they run (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the same bounds as icpx_bnb_eval_async
 * (bit for bit), enqueued on the context's stream, with no allocation of any kind inside the call. items_dev and
 * out_dev (2 * nitems doubles, [ub..., lb...]) are DEVICE memory of the context's GPU; the per-slice partial sums
 * go to a fixed 1 MiB area inside the library's code object instead of a grown buffer, which limits a call to
 * ICPX_NOALLOC_MAX_ITEMS items (ICPX_ELIMIT beyond). An item whose level is >= ICPX_MAXROTLEVEL gets NaN bounds
 * (the host cannot inspect device-resident items). All calls on one GPU share that area, whichever context they
 * come through: calls that overlap in time on one GPU (different streams or contexts) race; calls on one stream
 * are ordered and safe. Requires the EDT and the data cloud (icpx_goicp_set_data) to be in place already. */
#define ICPX_NOALLOC_MAX_ITEMS 4096
int icpx_bnb_eval_noalloc(icpx_ctx* ctx, const icpx_bnb_item* items_dev, size_t nitems, double* out_dev);

/* Replaces the SSE loop of GoIcp::runICP (src/GoIcp.cpp:113-135) and the initial error of OuterBnB
 * (:161-172): d_i = Distance((float)(T * [x_i;1])) for the data cloud, T row-major 4x4 double (host);
 * writes every d_i (n floats, host, caller's order) so the caller can form the sequential float sum.
 * Distance here is LDT::Distance on the context's distance transform. The committed reference instantiates its `dt`
 * member as a KDTree (src/GoIcp.cpp:76-77), i.e. sums exact nearest-neighbour distances: icpx_nn_distance on the
 * transformed points gives those (icp::GoIcp::exact_residuals does that). */
int icpx_goicp_residuals(icpx_ctx* ctx, const double T[16], float* dist_out);

#ifdef __cplusplus
}
#endif
#endif /* ICPX_H */
